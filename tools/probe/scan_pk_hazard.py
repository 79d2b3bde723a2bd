"""Scan the gfx950 ISA of csrc/*.hip for the hazard found in round 4: a packed-fp32 VOP3P instruction
(v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: two passes) whose result register is read by an LDS or vector-memory instruction
in the VERY NEXT issue slot.  hipcc (ROCm 7.2) pads a dependent VALU consumer with `s_nop 0` but not a DS / VMEM consumer; on
MI355X the ds_bpermute_b32 of the LayerNorm backward's cross-lane sum read a stale register in ~1 of 10^6 waves when it
overlapped another kernel (tools/probe/merge_bisect.py).

    python tools/probe/scan_pk_hazard.py [-D...]        exit status 1 if any site is found
"""
import os, re, subprocess, sys
from concurrent.futures import ThreadPoolExecutor

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(REPO, "vision-transformers-pytorch_amd", "csrc")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "vision-transformers-pytorch_amd"))
from vtx.build import FLAGS   # the flags of the shipped library: a scan validates THAT binary
MEM = ("ds_", "global_", "buffer_", "flat_", "scratch_")


def regs(tok):
    """VGPR numbers named by an operand token: v12 -> {12}, v[4:7] -> {4..7}."""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def scan(path, extra):
    out = f"/tmp/scan_{os.path.basename(path)}.s"
    r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + ["-S", "--cuda-device-only", "-o", out, path], capture_output=True, text=True)
    if r.returncode:
        return [(path, 0, "COMPILE FAILED: " + r.stderr[-300:], "")]
    hits, kernel, prev = [], "?", None
    for ln, line in enumerate(open(out), 1):
        t = line.strip()
        if t.endswith(":") and not t.startswith("."):
            kernel = t[:-1]
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        op = t.split()[0]
        ops = [o.strip() for o in t[len(op):].split(";")[0].split(",")]
        if prev is not None and op.startswith(MEM):
            dst = prev[1]
            # every operand of a memory instruction that names a VGPR is either read (address, data) or written (load result):
            # loads write their first operand, everything else is read
            reads = ops[1:] if ("load" in op or "read" in op or "bpermute" in op or "permute" in op or "swizzle" in op or "atomic" in op and "_rtn" in op) else ops
            if "bpermute" in op or "permute" in op or "swizzle" in op:
                reads = ops[1:]
            rd = set().union(*[regs(o.split()[0]) for o in reads if o]) if reads else set()
            if rd & dst:
                hits.append((os.path.basename(path), ln, f"{prev[0]}  ->  {t.split(';')[0].strip()}", kernel[:90]))
        prev = (t.split(";")[0].strip(), set().union(*[regs(ops[0].split()[0])] if ops and ops[0] else [set()])) if op.startswith("v_pk_") and op.endswith("_f32") else None
    return hits


def main():
    extra = [a for a in sys.argv[1:] if a.startswith("-")]
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    with ThreadPoolExecutor(8) as ex:
        res = list(ex.map(lambda f: scan(f, extra), files))
    n = spills = 0
    for hits in res:
        for f, ln, what, kern in hits:
            if "scratch_store" in what:          # a register spill the compiler placed there itself (fp32 parity-mode kernels): reported, not failed
                spills += 1
                continue
            n += 1
            print(f"{f}:{ln}: {what}    [{kern}]")
    print(f"{n} packed-fp32 -> LDS / vector-memory back-to-back read sites in {len(files)} files (+ {spills} compiler spill stores)")
    return 1 if n else 0


if __name__ == "__main__":
    sys.exit(main())

"""Scan the gfx950 ISA of csrc/*.hip for the hazard found in round 5 (profiles/round5_mfma_branch_hazard.md): the result of an MFMA
read -- v_accvgpr_read_b32 or any VALU / memory instruction naming the destination registers -- on the TAKEN path of a branch
that follows the MFMA by a few instructions.  hipcc (ROCm 7.2) pads MFMA -> read with the required wait states in straight-line
code, but not when the read sits at the head of the branch target: a chain of v_mfma_f32_16x16x4_f32, `s_and_saveexec` /
`s_cbranch_execz` around the predicated load of the next operand tile, then `v_accvgpr_read` three instructions after the last MFMA
returned the accumulator before the chain had drained (fp32 attention, 10 key tiles).

    python tools/r5/scan_mfma_branch.py [file.s ...]        exit status 1 if any site is found
Without arguments: compiles csrc/*.hip with the library's flags (or reuses /tmp/scan_<file>.s of tools/probe/scan_pk_hazard.py
when newer than the sources)."""
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(REPO, "vision-transformers-pytorch_amd", "csrc")
sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))
from vtx.build import FLAGS   # the flags of the shipped library: a scan validates THAT binary

LOOK_BACK, LOOK_AHEAD = 12, 12
MIN_GAP = 6           # issue slots between the MFMA and the first reader on the taken path (s_nop N counts N + 1): the proven failure
                      # is 3 (fp32 16x16x4 chains); sites with 6-9 slots (bf16 16x16x32, whole-suite parity green) are counted, not failed
MIN_GAP_F32 = 11      # v_mfma_f32_16x16x4_f32 is an 8-pass instruction (32 cycles): its readers need 11 wait states
NEAR_GAP = 12


def regs(tok):
    """register numbers named by an operand token, as ('v' | 'a', n) pairs"""
    tok = tok.strip()
    m = re.fullmatch(r"([va])(\d+)", tok)
    if m:
        return {(m.group(1), int(m.group(2)))}
    m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", tok)
    if m:
        return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    return set()


def parse(path):
    """-> {kernel: [(label | None, op, [operands], text)]}"""
    kernels, cur, name = {}, None, None
    pending = []
    for line in open(path):
        t = line.split(";")[0].strip()
        if not t or t.startswith("//") or (t.startswith(".") and not t.endswith(":")):
            continue
        if t.endswith(":"):
            lab = t[:-1]
            if not lab.startswith(".L"):
                name, cur = lab, []
                kernels[name] = cur
            else:
                pending.append(lab)
            continue
        if cur is None:
            continue
        op = t.split()[0]
        ops = [o.strip() for o in t[len(op):].split(",")] if len(t) > len(op) else []
        cur.append((tuple(pending), op, ops, t))
        pending = []
    return kernels


def slots(ins):
    op, ops = ins[1], ins[2]
    if op == "s_nop":
        try:
            return int(ops[0]) + 1
        except Exception:
            return 1
    return 1


def scan_file(path):
    hits = []
    for kern, code in parse(path).items():
        labels = {}
        for i, ins in enumerate(code):
            for lab in ins[0]:
                labels[lab] = i
        for i, ins in enumerate(code):
            if not ins[1].startswith("s_cbranch"):
                continue
            tgt = labels.get(ins[2][0]) if ins[2] else None
            if tgt is None:
                continue
            # the last MFMA before the branch
            back, j = 0, i - 1
            while j >= 0 and back < LOOK_BACK and not code[j][1].startswith("v_mfma") and not code[j][1].startswith("v_smfma"):
                back += slots(code[j])
                j -= 1
            if j < 0 or back >= LOOK_BACK or not code[j][1].startswith(("v_mfma", "v_smfma")):
                continue
            dst = regs(code[j][2][0]) if code[j][2] else set()
            ahead, k = 0, tgt
            while k < len(code) and ahead < LOOK_AHEAD:
                op, ops = code[k][1], code[k][2]
                if op.startswith(("v_mfma", "v_smfma")):
                    # MFMA -> MFMA on the same accumulator is interlocked; an MFMA reading dst as A / B is not expected here
                    rd = set().union(*[regs(o) for o in ops[1:3]]) if len(ops) >= 3 else set()
                else:
                    reads = ops[1:] if (op.startswith("v_") or "load" in op or "read" in op) else ops
                    rd = set().union(*[regs(o.split()[0]) for o in reads if o]) if reads else set()
                if rd & dst:
                    gap = back + 1 + ahead          # (+1: the branch itself)
                    if gap < NEAR_GAP:
                        hits.append((os.path.basename(path), kern[:100], code[j][3], ins[3], code[k][3], gap))
                    break
                if op.startswith(("s_cbranch", "s_branch", "s_endpgm")):
                    break
                ahead += slots(code[k])
                k += 1
    return hits


def isa_of(src):
    out = f"/tmp/scan_{os.path.basename(src)}.s"
    deps = [src] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    if os.path.exists(out) and os.path.getmtime(out) > max(os.path.getmtime(d) for d in deps):
        return out
    r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-S", "--cuda-device-only", "-o", out, src], capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError(f"hipcc failed for {src}: {r.stderr[-300:]}")
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("-")]
    if args:
        files = args
    else:
        srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
        with ThreadPoolExecutor(8) as ex:
            files = list(ex.map(isa_of, srcs))
    n = near = 0
    for f in files:
        for fn, kern, mfma, br, rd, gap in scan_file(f):
            if gap >= (MIN_GAP_F32 if "16x16x4_f32" in mfma else MIN_GAP):
                near += 1
                continue
            n += 1
            print(f"{fn}: [{kern}]\n    {mfma}\n    {br}  (taken)\n    {rd}    <- {gap} issue slots after the MFMA")
    print(f"{n} MFMA results read at the head of a branch target within {MIN_GAP} issue slots ({MIN_GAP_F32} for the 8-pass fp32 MFMA; {near} more within {NEAR_GAP}), {len(files)} files")
    return 1 if n else 0


if __name__ == "__main__":
    sys.exit(main())

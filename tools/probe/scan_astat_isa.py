"""Compile-time invariants of the A-stationary GEMM (csrc/gemm_astat.hip) that its hand-counted waits rely on, checked on the gfx950
assembly hipcc generates (no GPU needed).  Every gemm_astat_kernel instantiation must have

  1. no register spills (scratch / buffer private-segment traffic goes through vmcnt: the counted s_waitcnt vmcnt(N) would be off);
  2. no flat_* instruction (a flat access counts in vmcnt AND lgkmcnt; hipcc emits one for a volatile / generic LDS pointer);
  3. in the kernels that read a residual / z vector per output vector (RESID, act'): the reserved landing registers v152 .. v167 and
     every AGPR untouched outside the inline-asm blocks (the vectors land there asynchronously: a compiler temporary in one of them
     would be overwritten).  amdgpu_num_vgpr(152) is a target, not a fence: the GELU-forward instantiations do go past it (they have
     no landing registers in use), so this is checked, not assumed;
  4. no compiler-inserted `s_waitcnt vmcnt(0)` between the first and the last s_barrier of the kernel, i.e. inside the k-step loops
     of either wave role (that wait is the acknowledgement of every store the wave has issued: what v3 of the kernel exists to avoid;
     hipcc inserts it in front of LDS reads of a wave that issued LDS-DMA, in front of LDS atomics, and for loop-carried loads).

    python tools/probe/scan_astat_isa.py        exit status 1 on a violation
"""
import os, re, subprocess, sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(REPO, "vision-transformers-pytorch_amd", "csrc", "gemm_astat.hip")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "vision-transformers-pytorch_amd"))
from vtx.build import FLAGS   # the flags of the shipped library: a scan validates THAT binary
RESERVED0 = 224 if any(a.startswith("-DAS_NCWD=4") for a in sys.argv[1:]) else 152


def vgprs(line):
    out = []
    for m in re.finditer(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]", line):
        out.append(int(m.group(1)) if m.group(1) else int(m.group(3)))
    return out


def main():
    out = "/tmp/scan_gemm_astat.s"
    r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + [a for a in sys.argv[1:] if a.startswith("-")] + ["-S", "--cuda-device-only", "-o", out, SRC],
                       capture_output=True, text=True)
    if r.returncode:
        print("COMPILE FAILED:", r.stderr[-500:])
        return 1
    txt = open(out).read()
    bad = n = 0
    for m in re.finditer(r"^(_Z17gemm_astat_kernel\w+):[^\n]*\n(.*?)s_endpgm", txt, re.S | re.M):
        name, body = m.group(1)[22:46], m.group(2).split("\n")
        n += 1
        ta = re.match(r"ILi(\d)ELb([01])ELi(\d)ELb([01])ELb([01])E", name)
        vec = ta is not None and (ta.group(4) == "1" or ta.group(3) in ("2", "4"))
        barriers = [i for i, l in enumerate(body) if l.strip().startswith("s_barrier")]
        in_asm = False
        hi = 0
        for i, l in enumerate(body):
            t = l.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not t or t.startswith((";", ".")) or t.endswith(":"):
                continue
            code = t.split(";")[0]
            if re.match(r"(scratch_|buffer_(load|store))", code):
                print(f"{name}: spill / private-segment access: {code}"); bad += 1
            if code.startswith("flat_"):
                print(f"{name}: flat access: {code}"); bad += 1
            if not in_asm:
                if re.search(r"\ba(\d+)\b|\ba\[\d+:\d+\]", code) and "v_accvgpr" in code:
                    print(f"{name}: AGPR use outside inline asm: {code}"); bad += 1
                v = vgprs(code)
                if v:
                    hi = max(hi, max(v))
                if vec and v and max(v) >= RESERVED0:
                    print(f"{name}: reserved register used by the compiler: {code}"); bad += 1
                if barriers and barriers[0] < i < barriers[-1] and re.match(r"s_waitcnt\b.*vmcnt\(0\)", code):
                    print(f"{name}: compiler-inserted vmcnt(0) inside the loops (line {i}): {code}"); bad += 1
        if len(barriers) < 6:
            print(f"{name}: only {len(barriers)} s_barrier found -- the scanner no longer recognises the kernel"); bad += 1
    print(f"{bad} violations in {n} gemm_astat_kernel instantiations")
    return 1 if (bad or n == 0) else 0


if __name__ == "__main__":
    sys.exit(main())

"""Compile-time invariants of the wide-tile weight-gradient kernel (wgrad_wide_kernel, csrc/gemm_wgrad_glds.hip) on the gfx950
assembly hipcc generates (no GPU needed):

  1. no register spills / private-segment traffic (one workgroup of 12 waves per CU: 168 registers per wave, and a spill goes
     through vmcnt, which the request waves count by hand);
  2. no flat_* instruction;
  3. no compiler-inserted `s_waitcnt vmcnt(0)` between the first s_barrier and the last MFMA, i.e. in the loops (the hand-written waits sit in inline-asm
     blocks and are not counted): hipcc inserts one in front of an LDS read of a wave that issued LDS-DMA when it cannot prove the
     two disjoint -- the request waves read the liveness / row tables of the same LDS allocation;
  4. every transpose read and every MFMA of the k-loop is there (2 (4 + J) ds_read_b64_tr_b16 and 4 J v_mfma per k-step of a
     128 x 64 J tile: 20 / 24 for the 384-column tiles).

    python tools/probe/scan_wgrad_isa.py        exit status 1 on a violation
"""
import os, re, subprocess, sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(REPO, "vision-transformers-pytorch_amd", "csrc", "gemm_wgrad_glds.hip")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "vision-transformers-pytorch_amd"))
from vtx.build import FLAGS   # the flags of the shipped library: a scan validates THAT binary


def main():
    out = "/tmp/scan_wgrad_wide.s"
    r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-S", "--cuda-device-only", "-o", out, SRC], capture_output=True, text=True)
    if r.returncode:
        print("COMPILE FAILED:", r.stderr[-500:])
        return 1
    txt = open(out).read()
    bad = n = 0
    for m in re.finditer(r"^(_Z17wgrad_wide_kernel\w+):[^\n]*\n(.*?)s_endpgm", txt, re.S | re.M):
        name, body = m.group(1), m.group(2).split("\n")
        n += 1
        barriers = [i for i, l in enumerate(body) if l.strip().startswith("s_barrier")]
        mfmas = [i for i, l in enumerate(body) if l.strip().startswith("v_mfma_f32_16x16x32")]
        loop_end = mfmas[-1] if mfmas else 0        # the multiplying waves' k-loop is the last loop of the kernel (the epilogue's own
        #                                             __syncthreads, behind it, may drain whatever it likes)
        J = int(re.search(r"Li(\d)EE", name).group(1))    # 16-column accumulator tiles per wave: 128 x 64 J output tiles
        in_asm = False
        ntr = nmfma = 0
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
            ntr += code.startswith("ds_read_b64_tr_b16")
            nmfma += code.startswith("v_mfma_f32_16x16x32")
            if not in_asm and barriers and barriers[0] < i < loop_end and re.match(r"s_waitcnt\b.*vmcnt\(0\)", code):
                print(f"{name}: compiler-inserted vmcnt(0) inside the loops (line {i}): {code}"); bad += 1
        if len(barriers) < 4:
            print(f"{name}: only {len(barriers)} s_barrier found -- the scanner no longer recognises the kernel"); bad += 1
        if ntr != 2 * (4 + J) or nmfma != 4 * J:
            print(f"{name}: {ntr} transpose reads / {nmfma} MFMAs in the body (expected {2 * (4 + J)} / {4 * J}: one k-step, not unrolled)"); bad += 1
    print(f"{bad} violations in {n} wgrad_wide_kernel instantiations")
    # J = 6: lockstep and two-group loops (option WGRAD_WIDE = 1 | 2), each plain and row-mapped; J = 5, 4: lockstep, plain and row-mapped;
    # J = 3: lockstep, plain
    return 1 if (bad or n != 9) else 0


if __name__ == "__main__":
    sys.exit(main())

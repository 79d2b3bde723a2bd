"""Highest compiler-allocated VGPR per gemm_astat kernel (the reserved landing registers v152.. are excluded).  usage: python tools/r4/vgpr_highwater.py file.s"""
import re, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r"^(_Z17gemm_astat_kernel\w+):[^\n]*\n(.*?)s_endpgm", txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    hi = 0
    for r in re.finditer(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]", body):
        n = int(r.group(1)) if r.group(1) else int(r.group(3))
        if n < 152: hi = max(hi, n)
    sp = len(re.findall(r"scratch_|buffer_store_dword|buffer_load_dword", body))
    print(f"{hi + 1:4d} vgprs  scratch-ops {sp:3d}  {name[22:56]}")

"""Compile-time invariants of the fused MLP kernels (csrc/mlp_fused.hip) on the gfx950 assembly hipcc generates (no GPU needed).
VERDICT r5: the default forward `mlp_fwd_kernel<3, 12, false, false>` carried a 1-register spill (12 waves per workgroup: 168
registers per lane) that no scanner saw.  Checked for every instantiation the DISPATCHER can reach by default
(MF_FWD_DEFAULT = 12, MF_BWD_DEFAULT = 6 and its ff = 32 x odd fallback 4, each with and without the LayerNorm-backward fold of
round 6; C = 96 and C = 64):

  1. no scratch (`.amdhsa_private_segment_fixed_size 0`, no scratch_* / buffer_* private-segment instruction);
  2. no flat_* instruction;
  3. the register allocation admits the waves the launch bound asks for (12 waves: <= 168; 4 waves: <= 512);
  4. the hidden loop is there: the kernels hold v_mfma_f32_16x16x32_bf16 and (backward) ds_read_b64_tr_b16.

Other variant codes (tests / timing probes: 16 waves, z / h outputs at 12 waves) are listed with their scratch bytes, not judged.

    python tools/probe/scan_mlp_isa.py        exit status 1 on a violation
"""
import os, re, subprocess, sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(REPO, "vision-transformers-pytorch_amd", "csrc", "mlp_fused.hip")
sys.path.insert(0, os.path.join(REPO, "vision-transformers-pytorch_amd"))
from vtx.build import FLAGS   # the flags of the shipped library: a scan validates THAT binary

# (kernel, KS, WAVES, rest of the template list as mangled by hipcc)
DEFAULTS = [("mlp_fwd_kernel", ks, 12, f"Lb0ELb0ELb{lnf}EE") for ks in (2, 3) for lnf in (0, 1)] + \
           [("mlp_bwd_kernel", ks, 4, f"Lb1ELb0ELi0ELb{pair}ELb{lnb}EE") for ks in (2, 3) for pair in (1, 0) for lnb in (0, 1)]
# (forward: plain | norm_ff on its row operands, option LN_FOLD bit 2;  backward: paired stores | the ff = 32 x odd fallback, each plain | with the LayerNorm backward folded into the epilogue: option LN_FOLD)


def main():
    out = "/tmp/scan_mlp_fused.s"
    r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-S", "--cuda-device-only", "-o", out, SRC], capture_output=True, text=True)
    if r.returncode:
        print("COMPILE FAILED:", r.stderr[-500:])
        return 1
    txt = open(out).read()
    meta = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
        g = lambda k: int(re.search(k + r"\s+(\d+)", m.group(2)).group(1))
        meta[m.group(1)] = (g(r"\.amdhsa_next_free_vgpr"), g(r"\.amdhsa_private_segment_fixed_size"))
    bodies = {m.group(1): m.group(2) for m in re.finditer(r"^(_ZN\S*mlp_(?:fwd|bwd)_kernel\w+):[^\n]*\n(.*?)s_endpgm", txt, re.S | re.M)}
    bad = n = 0
    for kern, ks, waves, rest in DEFAULTS:
        pat = f"{kern}ILi{ks}ELi{waves}E{rest}"
        names = [k for k in meta if pat in k]
        if len(names) != 1 or names[0] not in bodies:
            print(f"{pat}: instantiation not found -- the scanner no longer recognises the kernel"); bad += 1
            continue
        name = names[0]
        n += 1
        vgpr, scratch = meta[name]
        alloc = (vgpr + 7) // 8 * 8
        budget = 512 // ((waves + 3) // 4) // 8 * 8
        if scratch:
            print(f"{pat}: {scratch} bytes of scratch per lane"); bad += 1
        if alloc > budget:
            print(f"{pat}: {alloc} registers allocated, {budget} admit {waves} waves per CU"); bad += 1
        nm = ntr = 0
        for l in bodies[name].split("\n"):
            code = l.strip().split(";")[0]
            if re.match(r"(scratch_|buffer_(load|store))", code):
                print(f"{pat}: spill / private-segment access: {code}"); bad += 1
            if code.startswith("flat_"):
                print(f"{pat}: flat access: {code}"); bad += 1
            nm += code.startswith("v_mfma_f32_16x16x32")
            ntr += code.startswith("ds_read_b64_tr_b16")
        if nm < 8 * ks or (kern == "mlp_bwd_kernel" and ntr == 0):
            print(f"{pat}: {nm} MFMAs / {ntr} transpose reads in the body -- the hidden loop is not what the scanner expects"); bad += 1
        print(f"  {pat}: {vgpr} registers ({alloc} allocated of {budget}), scratch {scratch}, {nm} MFMAs, {ntr} transpose reads")
    others = sorted((k, v) for k, v in meta.items() if "mlp_" in k and v[1])
    for k, (vg, sc) in others:
        if not any(f"{kern}ILi{ks}ELi{w}E{rest}" in k for kern, ks, w, rest in DEFAULTS):
            print(f"  (not a default) {k[18:60]}...: {vg} registers, scratch {sc}")
    print(f"{bad} violations in {n} default fused-MLP instantiations")
    return 1 if (bad or n != len(DEFAULTS)) else 0


if __name__ == "__main__":
    sys.exit(main())

#!/bin/bash
# Copy the outputs of `tools/battery.sh prof TAG` (+ `bench`) from gpurun_out/ into the tracked profiles/roundN_* files.
#   tools/collect_profiles.sh TAG ROUND        e.g.  tools/collect_profiles.sh r2b 2
TAG=${1:?tag}; RN=${2:?round}
C=$(git rev-parse --short HEAD)
hdr() { echo "$1, commit $C (tools/battery.sh prof $TAG; single-stream VTX_SIDE_WGRAD=0 so that durations are attributable; 7 steps incl. warm-up; FillFunctor / copyBuffer rows are start-up work)."; echo; }
{ hdr "Swin-S B = 128 bf16, bench.py under rocprofv3 --kernel-trace --stats"; cat gpurun_out/prof_$TAG/kernel_stats.md; } > profiles/round${RN}_kernel_stats_swin_s_b128.md
{ hdr "ViT-S/16 B = 256 bf16"; cat gpurun_out/profvit_$TAG/kernel_stats.md; } > profiles/round${RN}_kernel_stats_vit_s16_b256.md
{ hdr "PVT-Small B = 128 bf16"; cat gpurun_out/profpvt_$TAG/kernel_stats.md; } > profiles/round${RN}_kernel_stats_pvt_small_b128.md
{ hdr "DINO DeiT-S/16 B = 64 (2 global + 8 local crops) bf16, 5 steps"; cat gpurun_out/profdino_$TAG/kernel_stats.md; } > profiles/round${RN}_kernel_stats_dino_deit_s16_b64.md
for m in swin_s vit_s16 pvt_small; do
  cp gpurun_out/pmc_traffic_$m.md profiles/round${RN}_pmc_traffic_$m.md
  python - "$m" "$RN" "$C" <<'PY'
import json, sys
m, rn, c = sys.argv[1:4]
d = json.load(open(f"gpurun_out/pmc_traffic_{m}.json"))
d["_meta"] = dict(d.get("_meta", {}), commit=c, round=int(rn))
json.dump(d, open(f"profiles/round{rn}_pmc_traffic_{m}.json", "w"), indent=1)
PY
done
for f in gemm_bench:gemm_microbench_vs_hipblaslt attn_bench:window_attention_microbench input_bench:input_pipeline_microbench hbm_floor:hbm_streaming_floor store_pattern:store_pattern_probe; do
  [ -s gpurun_out/${f%%:*}.log ] && cp gpurun_out/${f%%:*}.log profiles/round${RN}_${f##*:}.txt
done
for m in swin_s vit_s16 pvt_small dino twins_svt_s; do [ -s gpurun_out/bench_$m.log ] && cp gpurun_out/bench_$m.log profiles/round${RN}_bench_$m.json; done
ls -la profiles/ | grep round${RN}_

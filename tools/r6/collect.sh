#!/bin/bash
# Copy the outputs of `tools/r6/job.sh battery` / `prof <model>` from gpurun_out/ into the tracked profiles/round6_* files.
C=$(git rev-parse --short HEAD)
hdr() { echo "$1, bench.py under rocprofv3 --kernel-trace --stats, library of commit $C (tools/r6/job.sh prof; single-stream VTX_SIDE_WGRAD=0 so that durations are attributable; 7 steps incl. warm-up; FillFunctor / copyBuffer rows are start-up work)."; echo; }
{ hdr "Swin-S B = 128 bf16"; cat gpurun_out/r6_kernel_stats_swin_s.md; } > profiles/round6_kernel_stats_swin_s_b128.md
{ hdr "ViT-S/16 B = 256 bf16"; cat gpurun_out/r6_kernel_stats_vit_s16.md; } > profiles/round6_kernel_stats_vit_s16_b256.md
{ hdr "PVT-Small B = 128 bf16"; cat gpurun_out/r6_kernel_stats_pvt_small.md; } > profiles/round6_kernel_stats_pvt_small_b128.md
{ hdr "DINO DeiT-S/16 B = 64 (2 global + 8 local crops) bf16"; cat gpurun_out/r6_kernel_stats_dino.md; } > profiles/round6_kernel_stats_dino_deit_s16_b64.md
{ hdr "Twins-SVT-S B = 128 bf16"; cat gpurun_out/r6_kernel_stats_twins_svt_s.md; } > profiles/round6_kernel_stats_twins_svt_s_b128.md
for m in swin_s vit_s16 pvt_small; do
  cp gpurun_out/pmc_traffic_$m.md profiles/round6_pmc_traffic_$m.md
  python - "$m" "$C" <<'PY'
import json, sys
m, c = sys.argv[1:3]
d = json.load(open(f"gpurun_out/pmc_traffic_{m}.json"))
d["_meta"] = dict(d.get("_meta", {}), commit=c, round=6)
json.dump(d, open(f"profiles/round6_pmc_traffic_{m}.json", "w"), indent=1)
PY
  { echo "(library of commit $C: python bench.py --model $m --steps 20 --warmup 5 --shape-table; one event-sampled step behind the timed region)"; cat gpurun_out/r6_shape_table_$m.md; } > profiles/round6_shape_table_$m.md
done
cp gpurun_out/r6_gemm_bench.log profiles/round6_gemm_microbench_vs_hipblaslt.txt
cp gpurun_out/r6_attn_bench.log profiles/round6_window_attention_microbench.txt
for m in swin_s vit_s16 pvt_small dino twins_svt_s; do [ -s gpurun_out/r6_bench_$m.json ] && cp gpurun_out/r6_bench_$m.json profiles/round6_bench_$m.json; done
python tools/r6/excess.py profiles/round6_shape_table_swin_s.md 40 > profiles/round6_excess_over_floors_swin_s.md
cp gpurun_out/r6_host_ahead.log profiles/round6_host_ahead.txt
ls profiles | grep round6_

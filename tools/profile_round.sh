#!/bin/bash
# Regenerates the rocprofv3 evidence under gpurun_out/ on a GPU box (run from the repo root through gpurun):
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/profile_round.sh r06'
# then copy gpurun_out/<tag>_* into profiles/.  Kernel timing and the PMC passes are separate runs (rocprofv3 --pmc
# must not be combined with other trace domains on this pool).
set -u
TAG=${1:-r06}
R=$(pwd)
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# one context on one stream (--pipeline 1 --lanes 1) and no point-heads pass: with the default parallel lanes two kernels share the GPU, a
# kernel's duration in the trace (and its PMC counters) then include its neighbour - bench.py's own per-kernel
# figures (roofline.*) come from a single-stream pass (acrmi_profile_ops) as well
BENCH="python $R/bench.py --no-cpu-baseline --no-point-heads --no-latency --no-pmc --no-reduced-precision --lanes 1 --pipeline 1"
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/${TAG}_stats" -o bench -- $BENCH --steps 5 --warmup 2 > "$OUT/${TAG}_stats.log" 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d "$OUT/${TAG}_pmc_fetch" -o p -- $BENCH --steps 1 --warmup 1 > "$OUT/${TAG}_pmc_fetch.log" 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d "$OUT/${TAG}_pmc_write" -o p -- $BENCH --steps 1 --warmup 1 > "$OUT/${TAG}_pmc_write.log" 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 -d "$OUT/${TAG}_pmc_mfma" -o p -- $BENCH --steps 1 --warmup 1 > "$OUT/${TAG}_pmc_mfma.log" 2>&1
# the 16-bit program (reported next to the headline): kernel timing + HBM traffic of the f16 convolutions
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/${TAG}_fp16_stats" -o bench -- $BENCH --precision fp16 --steps 5 --warmup 2 > "$OUT/${TAG}_fp16_stats.log" 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d "$OUT/${TAG}_fp16_pmc_fetch" -o p -- $BENCH --precision fp16 --steps 1 --warmup 1 > "$OUT/${TAG}_fp16_pmc_fetch.log" 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d "$OUT/${TAG}_fp16_pmc_write" -o p -- $BENCH --precision fp16 --steps 1 --warmup 1 > "$OUT/${TAG}_fp16_pmc_write.log" 2>&1
# the split-operand program ('fp16x3': fp32 storage, conv_x3_kernel on the 16-bit matrix pipe): kernel timing + matrix-pipe occupancy
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/${TAG}_fp16x3_stats" -o bench -- $BENCH --precision fp16x3 --steps 5 --warmup 2 > "$OUT/${TAG}_fp16x3_stats.log" 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d "$OUT/${TAG}_fp16x3_pmc_mfma" -o p -- $BENCH --precision fp16x3 --steps 1 --warmup 1 > "$OUT/${TAG}_fp16x3_pmc_mfma.log" 2>&1
# the single-frame call (the reference's operating point, acr/main.py:126-141): kernel durations of batch-1 / batch-8 calls on
# a small-batch context (split-K lowering), lanes 1 / 2 / 4 (tools/latency_probe.py)
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/${TAG}_batch1_stats" -o b1 -- python $R/tools/latency_probe.py > "$OUT/${TAG}_batch1_stats.log" 2>&1
cd "$R"
cp "$(find "$OUT/${TAG}_batch1_stats" -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_batch1_kernel_stats.csv"
cp "$(find "$OUT/${TAG}_stats" -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_bench_kernel_stats.csv"
cp "$(find "$OUT/${TAG}_fp16_stats" -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_fp16_kernel_stats.csv"
cp "$(find "$OUT/${TAG}_fp16x3_stats" -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_fp16x3_kernel_stats.csv"
python tools/summarize_pmc.py "$OUT/${TAG}_pmc_fetch" "$OUT/${TAG}_pmc_write" "$OUT/${TAG}_hbm_traffic.json" > /dev/null
python tools/summarize_pmc.py "$OUT/${TAG}_fp16_pmc_fetch" "$OUT/${TAG}_fp16_pmc_write" "$OUT/${TAG}_fp16_hbm_traffic.json" > /dev/null
python tools/summarize_mfma.py "$OUT/${TAG}_pmc_mfma" "$OUT/${TAG}_pmc_mfma.txt"
python tools/summarize_mfma.py "$OUT/${TAG}_fp16_pmc_fetch" "$OUT/${TAG}_fp16_pmc_mfma.txt"
python tools/summarize_mfma.py "$OUT/${TAG}_fp16x3_pmc_mfma" "$OUT/${TAG}_fp16x3_pmc_mfma.txt"
tail -1 "$OUT/${TAG}_stats.log" | cut -c1-200
head -8 "$OUT/${TAG}_bench_kernel_stats.csv"
tail -1 "$OUT/${TAG}_fp16_stats.log" | cut -c1-200
head -6 "$OUT/${TAG}_fp16_kernel_stats.csv"
tail -1 "$OUT/${TAG}_fp16x3_stats.log" | cut -c1-200
head -6 "$OUT/${TAG}_fp16x3_kernel_stats.csv"

#!/bin/bash
# stage-alone timings (one stream) and counters of the fused walk on the atrium
set -o pipefail
mkdir -p gpurun_out/r3; export TMPDIR=/tmp
L=gpurun_out/r3/fused_diag.log
for f in 0 1; do
  echo "== STREAMS=1 FUSED=$f" | tee -a $L
  MI355PT_VERBOSE=1 MI355PT_STREAMS=1 MI355PT_FUSED=$f timeout -k 10 300 python scripts/perf_atrium.py 2>&1 | tee -a $L || exit 1
done
echo "== kernel trace, FUSED=1, two streams" | tee -a $L
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3/trace_C4 -o t -- python3 bench.py --config C4 --steps 1 --warmup 1 --spp 16 --no-cpu-baseline > gpurun_out/r3/trace_C4.log 2>&1 || { tail -5 gpurun_out/r3/trace_C4.log; exit 1; }
cp $(find gpurun_out/r3/trace_C4 -name '*kernel_stats.csv' | head -1) gpurun_out/r3/C4_fused_kernel_stats.csv; rm -rf gpurun_out/r3/trace_C4
head -8 gpurun_out/r3/C4_fused_kernel_stats.csv | tee -a $L
bash scripts/pmc_diag.sh r3 C4 8 2>&1 | tail -60 | tee -a $L

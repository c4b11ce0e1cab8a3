#!/bin/bash
# A/B of the fused walk (trace_fused.h) against the while-while kernels: parity tests first, then throughput of the atrium.
set -o pipefail
mkdir -p gpurun_out/r3
L=gpurun_out/r3/fused_ab.log; echo "==== $(date) $(git rev-parse --short HEAD 2>/dev/null)" >> $L
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "atrium or large_scene or tree_node or li_samples or film_vs or veach or intersection or bunny" > gpurun_out/r3/fused_tests.log 2>&1 || { tail -30 gpurun_out/r3/fused_tests.log; exit 1; }
tail -3 gpurun_out/r3/fused_tests.log
for s in 1 2; do for f in 0 1; do
  echo "== STREAMS=$s MI355PT_FUSED=$f" | tee -a $L
  MI355PT_STREAMS=$s MI355PT_FUSED=$f timeout -k 10 300 python scripts/perf_atrium.py 2>&1 | tee -a $L || exit 1
done; done
for thr in 32 40 56; do
  echo "== thr $thr" | tee -a $L
  MI355PT_FUSED_THR=$thr timeout -k 10 300 python scripts/perf_atrium.py 2>&1 | tee -a $L || exit 1
done
for g in 1536 2048 3072; do
  echo "== grid $g" | tee -a $L
  MI355PT_FUSED_GRID=$g timeout -k 10 300 python scripts/perf_atrium.py 2>&1 | tee -a $L || exit 1
done

#!/bin/bash
# parity (whole GPU suite) + throughput of the three single-GPU workloads
set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3/gpu_tests.log 2>&1 || { tail -30 gpurun_out/r3/gpu_tests.log; exit 1; }
tail -2 gpurun_out/r3/gpu_tests.log
for c in C2 C3 C4; do timeout -k 10 600 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:2], d['value'], 'Msamples/s', d['ms_per_step'], 'ms', d.get('roofline') and (d['roofline']['kernel'], d['roofline']['avg_launch_us'], d['roofline']['frac']), d.get('pipeline') and d['pipeline']['stage_ms_per_step'])" | tee -a gpurun_out/r3/quick.log || exit 1; done
for f in classic responsive; do timeout -k 10 600 python bench.py --face $f --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('face', d['config']['face'][:40], d['value'], 'Msamples/s', d['ms_per_step'], 'ms')" | tee -a gpurun_out/r3/quick.log || exit 1; done

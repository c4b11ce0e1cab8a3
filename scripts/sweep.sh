#!/bin/bash
# parameter sweep of the launch geometry on one workload: scripts/sweep.sh <config> (prints Msamples/s per setting)
C=${1:-C2}
run() { echo -n "$* : "; env "$@" timeout -k 10 300 python bench.py --config $C --steps 2 --warmup 1 --no-cpu-baseline --no-stage-timing | python -c "import sys, json; print(json.loads(sys.stdin.read())['value'])"; }
run MI355PT_DUMMY=0
for g in 1024 2048 8192 16384; do run MI355PT_GRID_EXTEND=$g; done
for g in 1024 2048 8192 16384; do run MI355PT_GRID_SHADOW=$g; done
for g in 256 384 768 1024; do run MI355PT_GRID_SHADE=$g; done
for s in 8192 32768 65536; do run MI355PT_SEGMENTS=$s; done
for b in 8388608 33554432; do run MI355PT_BATCH_PATHS=$b; done

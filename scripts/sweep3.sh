#!/bin/bash
run() { local c=$1; shift; echo -n "$c $* : "; env "$@" timeout -k 10 300 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-stage-timing 2>/dev/null | python -c "import sys, json; print(json.loads(sys.stdin.read())['value'])"; }
for rep in 1 2; do for g in 512 640 768 896 1024; do run C2 MI355PT_GRID_SHADE=$g; done; done
for g in 512 768 1024 1536; do run C4 MI355PT_GRID_SHADE=$g; done
for g in 512 768; do run C3 MI355PT_GRID_SHADE=$g; done

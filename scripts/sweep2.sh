#!/bin/bash
run() { local c=$1; shift; echo -n "$c $* : "; env "$@" timeout -k 10 300 python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-stage-timing 2>/dev/null | python -c "import sys, json; print(json.loads(sys.stdin.read())['value'])"; }
for c in C2 C3 C4; do
  for b in 16777216 33554432 50331648 67108864 134217728; do run $c MI355PT_BATCH_PATHS=$b; done
  run $c MI355PT_BATCH_PATHS=33554432 MI355PT_SEGMENTS=32768
  run $c MI355PT_BATCH_PATHS=67108864 MI355PT_SEGMENTS=32768
  run $c MI355PT_BATCH_PATHS=67108864 MI355PT_SEGMENTS=65536
done

#!/bin/bash
# A/B of the whole-table searches of emitter sampling (DScene::search_flags; MI355PT_SEARCH=0 switches them off).  Same box, alternating.
set -o pipefail
mkdir -p gpurun_out/r3
for rep in 1 2; do for f in auto 0; do for c in ${CFGS:-C4 C3 C2}; do
  env $([ $f = auto ] || echo MI355PT_SEARCH=$f) timeout -k 10 300 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('search=$f', d['config']['workload'][:2], d['value'], d['roofline']['avg_launch_us'], d['pipeline']['stage_ms_per_step']['shade_ms'])" | tee -a gpurun_out/r3/search_ab.log || exit 1
done; done; done

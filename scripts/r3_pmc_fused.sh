#!/bin/bash
# single-stream counters of the fused walk: instruction classes, issue / wait cycles, busy cycles; plus the list of SQ counters this rocprofv3 knows
set -o pipefail
export TMPDIR=/tmp MI355PT_STREAMS=1
OUT=gpurun_out/${TAG:-r3c}; mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ|TCP|TA|TCC|GRBM)_[A-Z0-9_a-z]+" | sort -u > $OUT/counters.txt; wc -l $OUT/counters.txt
run() { local name=$1; shift
  timeout -k 10 600 rocprofv3 --pmc "$@" --output-format csv -d $OUT/p_$name -o t -- python3 bench.py --config ${CFG:-C4} --steps 1 --warmup 0 --spp ${SPP:-8} --no-cpu-baseline --no-stage-timing > $OUT/p_$name.log 2>&1 \
    && cp $(find $OUT/p_$name -name '*counter_collection.csv' | head -1) $OUT/p_$name.csv || { echo "pass $name failed"; tail -5 $OUT/p_$name.log; }
  rm -rf $OUT/p_$name; }
run a SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_BRANCH
run b SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU
run c GRBM_GUI_ACTIVE
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 bench.py --config ${CFG:-C4} --steps 1 --warmup 0 --spp ${SPP:-8} --no-cpu-baseline --no-stage-timing > $OUT/trace.log 2>&1 && cp $(find $OUT/trace -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv; rm -rf $OUT/trace
python3 - <<PY
import pandas as pd, glob, re
for f in sorted(glob.glob("$OUT/p_*.csv")):
    d = pd.read_csv(f); d["k"] = d["Kernel_Name"].map(lambda s: (re.search(r"(k_[a-z_]+(<[^>]*>)?)", s) or [None, "other"])[1] if "k_" in s else "other")
    t = d.pivot_table(index="k", columns="Counter_Name", values="Counter_Value", aggfunc="mean")
    print("==", f); print(t.to_string(float_format=lambda v: "%.4g" % v))
PY
head -6 $OUT/kernel_stats.csv

#!/bin/bash
# Diagnostic counters of one workload's stage kernels, one rocprofv3 --pmc pass per hardware block (no trace domains alongside): instruction mix and wait
# cycles (SQ), texture-addresser / L1 activity (TA, TCP), L2 hits (TCC), address translation (UTCL1).  usage: scripts/pmc_diag.sh <round> <config> <spp>
R=${1:-r02}; C=${2:-C4}; SPP=${3:-8}; OUT=gpurun_out/$R; mkdir -p $OUT; export TMPDIR=/tmp
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -ge 124 ]; then echo "[pmc_diag] timed out / killed (rc $rc): stopping" >&2; exit $rc; fi; return $rc; }
pass() { local name=$1; shift
  run 600 rocprofv3 --pmc "$@" --output-format csv -d $OUT/diag_${C}_$name -o t -- python3 bench.py --config $C --steps 1 --warmup 0 --spp $SPP --no-cpu-baseline --no-stage-timing > $OUT/diag_${C}_$name.log 2>&1 \
    && cp $(find $OUT/diag_${C}_$name -name '*counter_collection.csv' | head -1) $OUT/diag_${C}_$name.csv || echo "pass $name failed (see $OUT/diag_${C}_$name.log)"
  rm -rf $OUT/diag_${C}_$name; }
pass sq SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
# TA block, ONE derived counter per pass: round 2's single pass of four TA counters was rejected at rocprofiler_create_counter_config ("error code 38: Request exceeds the
# capabilities of the hardware to collect", gpurun_out/r02.prev/diag_C4_ta.log) -- before the program launched anything: a counter-SET rejection (the TA block has two
# counter slots per instance and the *_sum / _avr forms take one each per shader engine), not a GPU fault.  Two counters per pass fit.
# what the waves wait FOR: vector loads vs stores vs LDS vs scalar memory (instruction counts and the cycles each kind keeps a wave busy), one group per pass (8 SQ counters fit)
pass sq2 SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_LDS
pass tcpw TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_GATE_EN1_sum
pass ta1 TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum
pass ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass tlb TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum
pass grbm GRBM_GUI_ACTIVE
python3 - <<PY
import pandas as pd, glob, re
for f in sorted(glob.glob("$OUT/diag_${C}_*.csv")):
    d = pd.read_csv(f); d["k"] = d["Kernel_Name"].map(lambda s: (re.search(r"(k_[a-z_]+(<[^>]*>)?)", s) or [None, "other"])[1] if "k_" in s else "other")
    t = d.pivot_table(index="k", columns="Counter_Name", values="Counter_Value", aggfunc="mean")
    print("==", f); print(t.to_string(float_format=lambda v: "%.4g" % v))
PY

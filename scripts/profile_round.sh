#!/bin/bash
# Measurement pass of one round on the GPU box (run through gpurun): for each BASELINE workload the unprofiled bench line, the rocprofv3 kernel trace of the
# same command, and the HBM traffic counters in their own passes (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass; never together with a trace).
# Everything lands in gpurun_out/<round>/; copy what is to be kept into profiles/ afterwards (scripts/collect_profiles.py does that).
# usage: scripts/profile_round.sh <round tag, e.g. r02> [configs, default "C2 C3 C4"]
R=${1:-r02}; CFGS=${2:-"C2 C3 C4"}; OUT=gpurun_out/$R; mkdir -p $OUT; export TMPDIR=/tmp
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -ge 124 ]; then echo "[profile_round] '$*' timed out / was killed (rc $rc): stopping" >&2; exit $rc; fi; return $rc; }
rocprofv3 -L > $OUT/counters_available.txt 2>&1
for C in $CFGS; do
  case $C in C2) SPP=64; BASE="";; C3) SPP=64; BASE="--no-cpu-baseline";; *) SPP=16; BASE="--no-cpu-baseline";; esac      # two full-size batches (one per path pool), as in the full job
  echo "== $C bench"; run 600 python3 bench.py --config $C $BASE > $OUT/${C}_bench.json 2> $OUT/${C}_bench.err || exit 1
  cat $OUT/${C}_bench.json | cut -c1-400
  echo "== $C kernel trace"; run 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${C}_trace -o t -- python3 bench.py --config $C --steps 2 --warmup 1 --no-cpu-baseline > $OUT/${C}_bench_under_rocprof.json 2> $OUT/${C}_trace.err || exit 1
  cp $(find $OUT/${C}_trace -name '*kernel_stats.csv' | head -1) $OUT/${C}_kernel_stats.csv
  for P in FETCH_SIZE WRITE_SIZE; do
    echo "== $C pmc $P"; run 900 rocprofv3 --pmc $P --output-format csv -d $OUT/${C}_pmc_$P -o t -- python3 bench.py --config $C --steps 1 --warmup 0 --spp $SPP --no-cpu-baseline --no-stage-timing > $OUT/${C}_pmc_$P.log 2>&1 || exit 1
    cp $(find $OUT/${C}_pmc_$P -name '*counter_collection.csv' | head -1) $OUT/${C}_pmc_$P.csv
  done
  python3 scripts/summarize_pmc.py $OUT/${C}_pmc_FETCH_SIZE.csv $OUT/${C}_pmc_WRITE_SIZE.csv $OUT/pmc_traffic.json $C | tee $OUT/${C}_pmc_summary.txt
  rm -rf $OUT/${C}_trace $OUT/${C}_pmc_FETCH_SIZE $OUT/${C}_pmc_WRITE_SIZE
done

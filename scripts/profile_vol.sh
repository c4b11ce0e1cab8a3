#!/bin/bash
# Kernel trace of the volumetric stages at 1080p: fog_box under volpath_simple and volpath, the fogged layered room (BSDF adapters), the fogged
# masked scene.  Run through gpurun; output in gpurun_out/<tag>_vol/, kept copies: profiles/<tag>_vol_*.  usage: scripts/profile_vol.sh <round tag>
R=${1:-r02}; OUT=gpurun_out/${R}_vol; mkdir -p $OUT; export TMPDIR=/tmp
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -ge 124 ]; then echo "[profile_vol] '$*' timed out / was killed (rc $rc): stopping" >&2; exit $rc; fi; return $rc; }
i=0
for ARGS in 'fog_box 1920 1080 32 {"global_fog":true}' 'fog_box 1920 1080 32 {"global_fog":true,"integrator":2}' 'layered_room 1920 1080 32 {"fog":2}' 'masked_room 1920 1080 32 {"fog":1}'; do
  i=$((i+1)); set -- $ARGS
  echo "== $ARGS"; run 300 python3 scripts/perf_scene.py "$@" | tee -a $OUT/rates.txt || exit 1
  run 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace$i -o t -- python3 scripts/perf_scene.py "$@" > $OUT/trace$i.log 2>&1 || exit 1
  cp $(find $OUT/trace$i -name '*kernel_stats.csv' | head -1) $OUT/vol${i}_kernel_stats.csv; rm -rf $OUT/trace$i
done

#!/bin/bash
# A/B an older checkout (_abA/) against the working tree on the same box
for i in 1 2; do
  echo -n "A atrium: "; python _abA/scripts/perf_atrium.py 2>&1 | tail -1 | cut -c44-200
  echo -n "B atrium: "; python scripts/perf_atrium.py 2>&1 | tail -1 | cut -c44-200
  echo -n "A veach:  "; python _abA/scripts/perf_veach.py 2>&1 | tail -1
  echo -n "B veach:  "; python scripts/perf_veach.py 2>&1 | tail -1
done

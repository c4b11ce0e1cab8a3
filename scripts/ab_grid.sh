#!/bin/bash
A=$1; B=$2
for g in 512 768 1024 1536; do
  echo -n "A g=$g: "; MI355PT_GRID_SHADE=$g MI355PT_LIB=$A python scripts/perf_quick.py 256 | cut -c30-140
  echo -n "B g=$g: "; MI355PT_GRID_SHADE=$g MI355PT_LIB=$B python scripts/perf_quick.py 256 | cut -c30-140
done

#!/bin/bash
# A/B two builds of libmi355pt.so on the same box: scripts/ab.sh <libA> <libB> [spp]; alternates runs to cancel drift
A=$1; B=$2; SPP=${3:-256}
for i in 1 2 3; do
  echo -n "A: "; MI355PT_LIB=$A python scripts/perf_quick.py $SPP | cut -c1-140
  echo -n "B: "; MI355PT_LIB=$B python scripts/perf_quick.py $SPP | cut -c1-140
done

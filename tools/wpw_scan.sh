#!/bin/bash
# kernel time vs wavefronts-per-walker for small batches, all on one box
for n in 32 64 128 256 512 1024; do
  for w in 1 2 4; do
    MAGPROP_AMD_WPW=$w python bench.py --no-cpu-baseline --no-mcmc --nwalk $n --steps 100 2>/dev/null | \
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('n=$n wpw=$w kernel_ms', round(d['roofline']['kernel_ms_avg'],4))"
  done
done

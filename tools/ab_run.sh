#!/bin/bash
# Interleaved A/B timing of kernel builds on ONE box (boxes differ by several per cent, so builds must be
# compared inside the same gpurun call):  bash tools/ab_run.sh libA.so libB.so ...
for rep in 1 2 3; do
  for lib in "$@"; do
    MAGPROP_AMD_LIB=$PWD/$lib python bench.py --no-cpu-baseline --no-mcmc --no-extra --steps 300 2>/dev/null | \
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', 'n1024 kernel_ms', round(d['roofline']['kernel_ms_avg'],4), 'min', round(d['roofline']['kernel_ms_min'],4))"
  done
done
for lib in "$@"; do
  MAGPROP_AMD_LIB=$PWD/$lib python bench.py --no-cpu-baseline --no-mcmc --no-extra --nwalk 4096 --grb Classic --steps 100 2>/dev/null | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', 'n4096 kernel_ms', round(d['roofline']['kernel_ms_avg'],4), 'min', round(d['roofline']['kernel_ms_min'],4))"
done

#!/bin/bash
# Same-box A/B of kernel builds over the headline, a large batch and mode B:  bash tools/ab_full.sh libA.so libB.so ...
for rep in 1 2; do
  for lib in "$@"; do
    for args in "" "--nwalk 8192" "--curve" "--curve --nwalk 4096"; do
      MAGPROP_AMD_LIB=$PWD/$lib python bench.py --no-cpu-baseline --no-mcmc --no-extra --steps 200 $args 2>/dev/null | \
        python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', '[$args]', 'kernel_ms', round(d['roofline']['kernel_ms_avg'],4), 'M evals/s', round(d['value']/1e6,3))"
    done
  done
done

#!/bin/bash
# kernel time vs steps-per-lane variant (SPL=4: one wave per SIMD, 256-step tiles; SPL=2: two waves per SIMD) by batch size, one box
for n in 1024 1152 1280 1536 2048 3072 4096 8192; do
  for v in 4 2; do
    MAGPROP_AMD_SPL=$v python bench.py --no-cpu-baseline --no-mcmc --nwalk $n --steps 60 2>/dev/null | \
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('n=$n spl=$v kernel_ms', round(d['roofline']['kernel_ms_avg'],4), 'evals/s', round(d['kernel_evals_per_sec_per_gpu']))"
  done
done

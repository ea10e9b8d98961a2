#!/bin/bash
# Steps per lane (tile length) against batch size on one box: MAGPROP_AMD_SPL=2|4 forces the one-wavefront kernels.
# the MAGPROP_AMD_* overrides are honoured by the developer build only: make -C magprop_amd/csrc experiments
export MAGPROP_AMD_LIB=${MAGPROP_AMD_LIB:-$PWD/magprop_amd/libmagprop_amd_exp.so}
for n in 256 512 768 1024 1536 2048 4096; do
  for spl in 4 2; do
    MAGPROP_AMD_SPL=$spl python bench.py --no-cpu-baseline --no-mcmc --no-extra --nwalk $n --steps 100 --warmup 5 2>/dev/null | \
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('spl=$spl n=$n kernel_ms', round(d['roofline']['kernel_ms_avg'],4), 'Mevals/s', round(d['kernel_evals_per_sec_per_gpu']/1e6,3))"
  done
  python bench.py --no-cpu-baseline --no-mcmc --no-extra --nwalk $n --steps 100 --warmup 5 2>/dev/null | \
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('auto  n=$n kernel_ms', round(d['roofline']['kernel_ms_avg'],4), 'Mevals/s', round(d['kernel_evals_per_sec_per_gpu']/1e6,3), d['config']['kernel_variant'])"
done

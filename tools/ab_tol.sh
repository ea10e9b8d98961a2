#!/bin/bash
# Newton-sweep tolerance A/B on one box: kernel time near the truth / prior-wide / burnt-in, per batch size.
# the MAGPROP_AMD_* overrides are honoured by the developer build only: make -C magprop_amd/csrc experiments
export MAGPROP_AMD_LIB=${MAGPROP_AMD_LIB:-$PWD/magprop_amd/libmagprop_amd_exp.so}
for rep in 1 2; do
for n in 1024 4096 8192 512; do
  for tol in 1e-9 1e-7; do
    MAGPROP_AMD_SWEEP_TOL=$tol python bench.py --no-cpu-baseline --no-mcmc --nwalk $n --steps 60 --warmup 5 2>/dev/null | \
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print('tol $tol n=$n', 'kernel_ms', round(d['roofline']['kernel_ms_avg'],4), 'near', round(k['near_truth']['kernel_ms'],4), k['near_truth']['sweeps_per_tile'], 'wide', round(k['prior_wide']['kernel_ms'],4), round(k['prior_wide']['sweeps_per_tile'],3), 'burnt', round(k['burnt_in_500_steps']['kernel_ms'],4), round(k['burnt_in_500_steps']['sweeps_per_tile'],3), 'golden', d['check']['golden']['max_rel_dev_vs_reference_tight_lsoda'], d['check']['golden']['pass'])"
  done
done
done

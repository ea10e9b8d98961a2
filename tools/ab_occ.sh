#!/bin/bash
# Occupancy experiment (same box, interleaved): resident waves per SIMD x steps per lane for batches beyond one wave per SIMD.
#   bash tools/ab_occ.sh   -> lines "lib spl nwalk kernel_ms evals/s"
run() {  # lib spl nwalk grb
  MAGPROP_AMD_LIB=$PWD/$1 MAGPROP_AMD_SPL=$2 python bench.py --no-cpu-baseline --no-mcmc --no-extra --nwalk $3 --grb $4 --steps 60 --warmup 5 2>/dev/null | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 spl=$2 n=$3', 'kernel_ms', round(d['roofline']['kernel_ms_avg'],4), 'Mevals/s', round(d['kernel_evals_per_sec_per_gpu']/1e6,3), 'lnprob0', d['check']['lnprob0'])"
}
for rep in 1 2; do
  for n in 8192 4096 2048 1536; do
    run ab_base.so 2 $n Humped
    run ab_s2w3.so 2 $n Humped
    run ab_s2w4.so 2 $n Humped
    run ab_s1w2.so 1 $n Humped
    run ab_s1w3.so 1 $n Humped
    run ab_s1w4.so 1 $n Humped
  done
done

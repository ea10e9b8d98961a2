#!/bin/bash
# Same-box A/B of kernel builds over the legs round 5 works on (small launches, headline, tails, sampler, 4 096 Classic, config 5):
#   bash tools/ab_round5.sh ab/libA.so ab/libB.so ...        (GPU box; boxes differ by several per cent, builds must share one)
REPS=${REPS:-2}
for rep in $(seq $REPS); do
  for lib in "$@"; do
    MAGPROP_AMD_LIB=$PWD/$lib python tools/ab_small.py $lib 2>/dev/null
    MAGPROP_AMD_LIB=$PWD/$lib python tools/ab_tail.py $lib 2>/dev/null | grep -v "^/opt"
    MAGPROP_AMD_LIB=$PWD/$lib python bench.py --no-cpu-baseline --no-extra --steps 200 --mcmc-steps 200 2>/dev/null | \
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['ensemble_sampler']; print('$lib', 'n1024 kernel_ms %.4f value %.3f M | sampler %.3f M w-steps/s (%.4f ms/step)' % (d['roofline']['kernel_ms_avg'], d['value']/1e6, s['walker_steps_per_sec']/1e6, s['ms_per_step']))"
    for args in "--nwalk 4096 --grb Classic" "--config 5"; do
      MAGPROP_AMD_LIB=$PWD/$lib python bench.py --no-cpu-baseline --no-mcmc --no-extra --steps 100 $args 2>/dev/null | \
        python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', '[$args]', 'kernel_ms %.4f value %.3f M' % (d['roofline']['kernel_ms_avg'], d['value']/1e6))"
    done
  done
done

#!/bin/bash
# Every kernel a BASELINE config dispatches to, one tools/profile.sh run each (GPU box).  bash tools/profile_all.sh <round>
R=${1:-r03}
PART=${2:-all}   # a | b | all (a gpurun call is limited to 20 minutes)
set -x
if [ $PART != b ]; then
KERN=lnprob_kernel    bash tools/profile.sh ${R}_n1024 --no-mcmc > /dev/null || exit 1
KERN=lnprob_kernel    bash tools/profile.sh ${R}_n4096_classic --config 3 --no-mcmc > /dev/null || exit 2
KERN=lnprob_kernel    bash tools/profile.sh ${R}_n8192 --config 4 --no-mcmc > /dev/null || exit 3
KERN=lnprob_team_kernel bash tools/profile.sh ${R}_n512 --nwalk 512 --no-mcmc > /dev/null || exit 6
KERN=lnprob_team_kernel bash tools/profile.sh ${R}_n256 --nwalk 256 --no-mcmc > /dev/null || exit 8
fi
if [ $PART = a ]; then echo done; exit 0; fi
KERN=lnprob_kernel    bash tools/profile.sh ${R}_curve1024 --curve --no-mcmc > /dev/null || exit 4
KERN=lnprob_kernel    bash tools/profile.sh ${R}_config5 --config 5 > /dev/null || exit 5
KERN=stretch_step_kernel bash tools/profile.sh ${R}_stretch_step1536 --steps 2 --warmup 1 --mcmc-steps 60 > /dev/null || exit 7
echo done

#!/bin/bash
# [KERN=<kernel name fragment>] tools/profile.sh <tag> [bench args...] — run ON THE GPU BOX (through gpurun).  Produces under gpurun_out/prof_<tag>/:
#   trace/  rocprofv3 --kernel-trace --stats of `bench.py`      (per-kernel time)
#   pmc_sq/ pmc_fetch/ pmc_write/  separate --pmc passes        (VALU issue mix; HBM bytes: MI355X_MICROARCH.md HBM section)
# Copy the summary written by tools/summarize_prof.py into profiles/ to have it judged.
set -o pipefail
TAG=${1:-run}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 30 --warmup 3 --no-cpu-baseline --no-extra $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.log || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -- python3 $REPO/bench.py $ARGS > $OUT/bench_pmc_sq.json 2> $OUT/pmc_sq.log || exit 2
rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py $ARGS > $OUT/bench_pmc_fetch.json 2> $OUT/pmc_fetch.log || exit 3
rocprofv3 --kernel-trace --pmc WRITE_SIZE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py $ARGS > $OUT/bench_pmc_write.json 2> $OUT/pmc_write.log || exit 4
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FLOPS_FP64 --output-format csv -d $OUT/pmc_mix -- python3 $REPO/bench.py $ARGS > $OUT/bench_pmc_mix.json 2> $OUT/pmc_mix.log || exit 5
rocprofv3 --kernel-trace --pmc SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_IFETCH SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_VALU_FLOPS_FP64_TRANS SQ_INSTS_VALU_INT64 --output-format csv -d $OUT/pmc_misc -- python3 $REPO/bench.py $ARGS > $OUT/bench_pmc_misc.json 2> $OUT/pmc_misc.log || exit 6
python3 $REPO/tools/summarize_prof.py $OUT $TAG ${KERN:-lnprob_kernel} > $OUT/summary.md
cat $OUT/summary.md

#!/bin/bash
# tools/pmc_extra.sh <tag> [bench args] -- GPU box: instruction-cache, scalar-cache and LDS-latency counters of the lnprob kernels
# (gpurun_out/pmc_<tag>/): SQ_INST_LEVEL_* / SQ_INSTS_* = average latency of an LDS / scalar-memory instruction in cycles.
set -o pipefail
TAG=${1:-run}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 30 --warmup 3 --no-cpu-baseline --no-extra --no-mcmc $@"
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVES SQ_WAVE_CYCLES --output-format csv -d $OUT/a -- python3 $REPO/bench.py $ARGS > $OUT/a.json 2> $OUT/a.log || exit 1
rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT/b -- python3 $REPO/bench.py $ARGS > $OUT/b.json 2> $OUT/b.log || exit 2
rocprofv3 --kernel-trace --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH --output-format csv -d $OUT/c -- python3 $REPO/bench.py $ARGS > $OUT/c.json 2> $OUT/c.log || exit 3
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(float); cnt = collections.defaultdict(int)
for f in glob.glob(out + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "lnprob" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
disp = max(cnt.values()) if cnt else 1
for k in sorted(tot):
    print(f"{k:32s} {tot[k] / cnt[k]:14.1f}   (avg per dispatch row, {cnt[k]} rows)")
PY

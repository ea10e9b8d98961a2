#!/bin/bash
# GPU box: the whole -m gpu suite, then the parity summaries it leaves under gpurun_out/ (hold-out set, soaks per seed).
#   bash tools/gpu_check.sh <tag>
TAG=${1:-check}
mkdir -p gpurun_out/$TAG
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/$TAG/tests.log 2>&1
echo "pytest rc $?"
tail -16 gpurun_out/$TAG/tests.log
python3 - <<'PY'
import glob, json
try:
    print("hold-out:", json.dumps(json.load(open("gpurun_out/holdout_parity.json"))))
except OSError:
    print("no hold-out summary")
for f in sorted(glob.glob("gpurun_out/soak_parity*seed*.json")):
    d = json.load(open(f))
    if "variants" in d:
        print(f, d["oracle_status_counts"], {k[:44]: (v["max_rel_diff"], v["p999_rel_diff"], v["status_mismatches"]) for k, v in d["variants"].items()})
    else:
        print(f, d)
PY

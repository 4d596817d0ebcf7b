#!/bin/bash
# tools/exp/pmc_logistic.sh LIB "CTRS": rocprofv3 --pmc CTRS over tools/exp/logistic_time.py; per-launch means of the logistic-path kernels
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
D=/tmp/pmcl_$$; rm -rf $D
rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $D -o p -- python3 $R/tools/exp/logistic_time.py $1 > /dev/null 2>&1
python3 - $D "$1" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for fn in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0][:40]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in agg:
    if "replay" in k or "logistic_dense" in k:
        print(sys.argv[2] or "product", k, {c: f"{agg[k][c] / n[k][c]:.4g}" for c in sorted(agg[k])})
PY
rm -rf $D

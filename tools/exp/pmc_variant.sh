#!/bin/bash
# tools/exp/pmc_variant.sh LIB DBG "CTRS": rocprofv3 --pmc CTRS over tools/exp/kernel_time.py practice62 LIB DBG; prints the per-launch mean of k_svr_dense
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
D=/tmp/pmc_$$; rm -rf $D
rocprofv3 --kernel-trace --pmc $3 --output-format csv -d $D -o p -- python3 $R/tools/exp/kernel_time.py practice62 $1 $2 > /dev/null 2>&1
python3 - $D "$1 $2" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(float); n = collections.Counter()
for fn in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "k_svr_dense" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
print(sys.argv[2], {k: round(agg[k] / n[k]) for k in sorted(agg)})
PY
rm -rf $D

R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
[ -x $R/tools/microbench/pmc_calib ] || hipcc --offload-arch=gfx950 -O2 -o $R/tools/microbench/pmc_calib $R/tools/microbench/pmc_calib.hip
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/cal_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/cal_$c -o p -- $R/tools/microbench/pmc_calib > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for c in ("FETCH_SIZE","WRITE_SIZE"):
    agg=collections.defaultdict(list)
    for fn in glob.glob(f"/tmp/cal_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"]==c: agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"])*1024)
    for k,v in sorted(agg.items()): print(c, k, [round(x/2**20,1) for x in v], "MiB; expected 512")
PY

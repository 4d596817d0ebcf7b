R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -E "WRREQ|WR_UNCACHED|EA0_WR|ATOMIC" | head -30
for c in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_WR_UNCACHED_32B_sum TCC_EA0_ATOMIC_sum"; do
  n=$(echo $c | tr ' ' '_')
  rm -rf /tmp/wr_$n /tmp/wc_$n
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/wr_$n -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-parity-gate > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/wc_$n -o p -- $R/tools/microbench/pmc_calib > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for tag in ("wr", "wc"):
    agg = collections.defaultdict(list)
    for fn in glob.glob(f"/tmp/{tag}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"].split("(")[0]
            if any(t in k for t in ("k_svr_dense", "k_records", "w8", "w16")):
                agg[(k[:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        print(tag, k, [round(x / 1e6, 3) for x in v[:3]], "M")
PY

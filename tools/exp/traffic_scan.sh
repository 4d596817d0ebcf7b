# FETCH_SIZE / WRITE_SIZE of k_svr_dense per launch against the number of regions (tiles) of the batch: is the excess over the algorithmic bytes
# (16 B per candidate: 8 read, 8 written) a per-launch constant, a per-tile one, or proportional?  (separate passes per counter)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for n in 1 4 16 62; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/ts_$c
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/ts_$c -o p -- python3 $R/tools/exp/kernel_time.py practice62:$n > /tmp/ts_$c.log 2>&1
  done
  python3 - $n <<'PY'
import csv, glob, sys, re
n = sys.argv[1]
cand = None
for line in open("/tmp/ts_FETCH_SIZE.log"):
    m = re.search(r"candidates (\d+)", line)
    if m: cand = int(m.group(1))
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v = {}
    for fn in glob.glob(f"/tmp/ts_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"] == c:
                k = r["Kernel_Name"].split("(")[0][:40]
                v.setdefault(k, []).append(float(r["Counter_Value"]) * 1024 * (2 if c == "FETCH_SIZE" else 1) / 1e6)
    out[c] = {k: round(sorted(x)[len(x) // 2], 2) for k, x in v.items()}
print(f"regions {n}: candidates {cand}, algorithmic {cand * 8 / 1e6:.2f} MB per direction; median MB per launch:")
for c in out:
    for k, x in out[c].items(): print(f"   {c} {k}: {x}")
PY
done

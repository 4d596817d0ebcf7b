# FETCH_SIZE / WRITE_SIZE of k_svr_dense per launch under ablations (scratch libraries of tools/exp/build_variant.sh):
#   product | nostore (-DSVR_NO_STORE: the score stores compiled out) | diag with the phase mask 7 (-DMIPGEN_DIAG: no scan, no tables, no candidate steps:
#   prologue + epilogue only) | diag 0 ; on the bench batch and on 2,048 exome exons
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
run() {   # config lib dbg
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/ta_$c
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/ta_$c -o p -- python3 $R/tools/exp/kernel_time.py $1 $2 $3 > /tmp/ta_$c.log 2>&1
  done
  python3 - "$1" "$2" "$3" <<'PY'
import csv, glob, sys, re
cand = None
for line in open("/tmp/ta_FETCH_SIZE.log"):
    m = re.search(r"candidates (\d+)", line)
    if m: cand = int(m.group(1))
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v = []
    for fn in glob.glob(f"/tmp/ta_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"] == c and "k_svr_dense" in r["Kernel_Name"]: v.append(float(r["Counter_Value"]) * 1024 * (2 if c == "FETCH_SIZE" else 1) / 1e6)
    res[c] = sorted(v)[len(v) // 2] if v else None
print(f"{sys.argv[1]:14s} {sys.argv[2] or 'product':44s} dbg {sys.argv[3] or '-':2s}: candidates {cand}, algorithmic {cand * 8 / 1e6:8.1f} MB per direction; FETCH {res['FETCH_SIZE']:8.1f} MB, WRITE {res['WRITE_SIZE']:8.1f} MB per launch (k_svr_dense, median)")
PY
}
for cfg in practice62 exome; do
  run $cfg "" ""
  run $cfg tools/exp/scratch/libmipgen_accel_nostore.so ""
  run $cfg tools/exp/scratch/libmipgen_accel_diag.so 0
  run $cfg tools/exp/scratch/libmipgen_accel_diag.so 7
done

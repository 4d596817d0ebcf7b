# FETCH_SIZE / WRITE_SIZE of kernels that touch no global memory (tools/microbench/ctx_probe.hip): is the per-launch constant of k_svr_dense under --pmc
# (~0.16 GB per direction) the counter collection's own wave-context traffic?  Separate passes per counter (MI355X_MICROARCH.md).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O2 -o /tmp/ctx_probe $R/tools/microbench/ctx_probe.hip || exit 1
echo "== without the profiler"; /tmp/ctx_probe
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/ctx_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/ctx_$c -o p -- /tmp/ctx_probe > /tmp/ctx_$c.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
names = ["A_1254wg_lds_long", "B_1254wg_lds_short", "C_1254wg_nolds_long", "D_1wg_lds_long"]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = []
    for fn in glob.glob(f"/tmp/ctx_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"] == c and "hold" in r["Kernel_Name"]:
                rows.append((int(r.get("Dispatch_Id", 0)), float(r["Counter_Value"])))
    rows.sort()
    # dispatch order = shapes A, B, C, D, three times over; FETCH_SIZE is doubled (the gfx950 correction), both are in KiB
    per = collections.defaultdict(list)
    for i, (_, v) in enumerate(rows): per[names[i % 4]].append(v * 1024 * (2 if c == "FETCH_SIZE" else 1) / 1e6)
    for n in names: print(c, n, [round(x, 1) for x in per[n]], "MB per launch (algorithmic: 0)")
PY

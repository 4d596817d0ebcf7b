#!/bin/bash
# tools/profile_round.sh TAG — the measurements behind DESIGN.md / bench.py's roofline object, on the MI355X box:
#   1. bench.py (default workload) -> gpurun_out/TAG_bench.json
#   2. rocprofv3 --kernel-trace --stats of the same command -> gpurun_out/TAG_kernel_stats.csv (+ the logistic / 5 kb / exome configs)
#   3. separate rocprofv3 --pmc passes (never combined with other trace domains) -> gpurun_out/TAG_pmc.json, TAG_hbm_traffic.json
# Run as:  gpurun --timeout 1800 -- 'bash tools/profile_round.sh r02d'   then copy the files into profiles/.
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 "$R/bench.py" --steps 20 --warmup 5 > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"
tail -c 400 "$OUT/${TAG}_bench.json"; echo
rm -rf "$OUT/prof_$TAG"
B="--steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-parity-gate"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$TAG/stats" -o s -- python3 "$R/bench.py" $B > /dev/null 2>&1
cp "$(find "$OUT/prof_$TAG/stats" -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_kernel_stats.csv"
for cfg in "regions5k --method logistic" "regions5k --method svr" "exome"; do
  name=$(echo $cfg | tr -d ' -' )
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$TAG/stats_$name" -o s -- python3 "$R/bench.py" --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-parity-gate > /dev/null 2>&1
  cp "$(find "$OUT/prof_$TAG/stats_$name" -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_kernel_stats_$name.csv"
done
i=0
for ctrs in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
            "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$OUT/prof_$TAG/pmc$i" -o p -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-parity-gate > /dev/null 2>&1
done
j=0
for ctrs in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  j=$((j+1))
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$OUT/prof_$TAG/lpmc$j" -o p -- python3 "$R/bench.py" --config regions5k --method logistic --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-parity-gate > /dev/null 2>&1
done
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, json, sys, collections
out, tag = sys.argv[1], sys.argv[2]
res = {}
for kern in ("k_svr_dense", "k_svr_finish", "k_records", "k_replay_condense"):
    agg = collections.defaultdict(float); n = collections.Counter()
    for fn in glob.glob(f"{out}/prof_{tag}/pmc*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if kern in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    res[kern] = {k: agg[k] / n[k] for k in sorted(agg)}
lres = {}
for kern in ("k_logistic_dense", "k_replay_condense", "k_collapse"):
    agg = collections.defaultdict(float); n = collections.Counter()
    for fn in glob.glob(f"{out}/prof_{tag}/lpmc*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if kern in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    lres[kern] = {k: agg[k] / n[k] for k in sorted(agg)}
lres["_note"] = ("mean per launch, summed over the device; separate --pmc passes of `bench.py --config regions5k --method logistic --steps 2 --warmup 1 "
                 "--no-cpu-baseline --no-extras` (24 regions of 5 kb, capture 120-250); FETCH_SIZE / WRITE_SIZE in KiB")
json.dump(lres, open(f"{out}/{tag}_pmc_logistic.json", "w"), indent=1)
res["_note"] = ("mean per launch, summed over the device; separate --pmc passes of `bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras` "
                "(default workload: practice62, capture 140-180, SVR n_sv=1024)")
json.dump(res, open(f"{out}/{tag}_pmc.json", "w"), indent=1)
k = res["k_svr_dense"]
if "FETCH_SIZE" in k and "WRITE_SIZE" in k:
    # counter unit: KiB.  FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950 (MI355X_MICROARCH.md); this kernel's reads are
    # narrow (8-byte records, SV rows served from L2), so the raw value is used.
    fetch, write = k["FETCH_SIZE"] * 1024.0, k["WRITE_SIZE"] * 1024.0
    json.dump({"k_svr_dense_bytes_per_launch": fetch + write, "fetch_bytes_raw": fetch, "write_bytes": write,
               "source": f"profiles/{tag}_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, tools/profile_round.sh; counter unit KiB)",
               "note": "writes: one 8-byte partial score per candidate and SV part, k_svr_finish (separate kernel) re-reads them"},
              open(f"{out}/{tag}_hbm_traffic.json", "w"), indent=1)
print(json.dumps(res["k_svr_dense"], indent=1))
PY
grep -E "k_svr_dense|k_records|k_replay|k_svr_finish|Name" "$OUT/${TAG}_kernel_stats.csv" | cut -c1-220

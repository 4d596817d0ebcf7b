#!/bin/bash
# tools/profile_round.sh TAG — the measurements behind DESIGN.md / bench.py's roofline object, on the MI355X box.  Everything is taken from
# the DRIVER's own command line (`bench.py --steps 20 --warmup 5`), so that profiles/ and BENCH_rNN.json can be recomputed from each other:
#   1. bench.py --steps 20 --warmup 5 --measure-traffic   -> gpurun_out/TAG_bench.json   (roofline.traffic measured in that run)
#   2. rocprofv3 --kernel-trace --stats of the same command (25 launches of every kernel; no parity gate / extras / CPU baseline in the
#      profiled process) -> gpurun_out/TAG_kernel_stats.csv (rocprofv3's own summary) and TAG_kernel_summary.json: per kernel the number of
#      calls, average, median, minimum and the average WITHOUT the first (cold) call, from the kernel trace itself
#   3. the same for the logistic / 5 kb / exome configs (--steps 5 --warmup 2)
#   4. separate rocprofv3 --pmc passes (never combined with other trace domains) -> gpurun_out/TAG_pmc.json, TAG_pmc_logistic.json,
#      TAG_hbm_traffic.json (FETCH_SIZE doubled: the gfx950 correction of MI355X_MICROARCH.md, HBM section; the raw value beside it)
# Run as:  gpurun --timeout 1800 -- 'bash tools/profile_round.sh r04a'   then copy the files into profiles/.
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 "$R/bench.py" --steps 20 --warmup 5 --measure-traffic > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"
tail -c 400 "$OUT/${TAG}_bench.json"; echo
rm -rf "$OUT/prof_$TAG"
B="--steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-parity-gate --no-measure-traffic"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$TAG/stats" -o s -- python3 "$R/bench.py" $B > /dev/null 2>&1
cp "$(find "$OUT/prof_$TAG/stats" -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_kernel_stats.csv"
for cfg in "regions5k --method logistic" "regions5k --method svr" "exome"; do
  name=$(echo $cfg | tr -d ' -' )
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$TAG/stats_$name" -o s -- python3 "$R/bench.py" --config $cfg --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-parity-gate --no-measure-traffic > /dev/null 2>&1
  cp "$(find "$OUT/prof_$TAG/stats_$name" -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_kernel_stats_$name.csv"
done
i=0
for ctrs in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
            "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$OUT/prof_$TAG/pmc$i" -o p -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-parity-gate --no-measure-traffic > /dev/null 2>&1
done
j=0
for ctrs in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  j=$((j+1))
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$OUT/prof_$TAG/lpmc$j" -o p -- python3 "$R/bench.py" --config regions5k --method logistic --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-parity-gate --no-measure-traffic > /dev/null 2>&1
done
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, json, sys, collections, statistics
out, tag = sys.argv[1], sys.argv[2]

def trace_summary(d):
    """per kernel: calls, average / median / min duration and the average without the first (cold) call, from the kernel TRACE (ns -> ms)"""
    per = collections.defaultdict(list)
    for fn in glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True):
        rows = sorted(csv.DictReader(open(fn)), key=lambda r: int(r["Start_Timestamp"]))
        for r in rows:
            per[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    res = {}
    for k, v in per.items():
        warm = v[1:] if len(v) > 1 else v
        res[k] = {"calls": len(v), "avg_ms": sum(v) / len(v), "median_ms": statistics.median(v), "min_ms": min(v), "max_ms": max(v),
                  "avg_ms_without_first_call": sum(warm) / len(warm), "first_call_ms": v[0]}
    return res

summ = {"_note": "rocprofv3 --kernel-trace of `bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-parity-gate` (the driver's line without the "
                 "untimed legs): 25 launches per kernel; durations in ms; the first call of a kernel is cold (code object load, tile lists): "
                 "avg_ms_without_first_call is the number to hold against BENCH's HIP-event kernel_ms"}
summ["default"] = trace_summary(f"{out}/prof_{tag}/stats")
for name in ("regions5kmethodlogistic", "regions5kmethodsvr", "exome"):
    summ[name] = trace_summary(f"{out}/prof_{tag}/stats_{name}")
json.dump(summ, open(f"{out}/{tag}_kernel_summary.json", "w"), indent=1)

def pmc(kerns, pat):
    res = {}
    for kern in kerns:
        agg = collections.defaultdict(float); n = collections.Counter()
        for fn in glob.glob(f"{out}/prof_{tag}/{pat}*/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(fn)):
                if kern in r["Kernel_Name"]:
                    agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
        if agg:                                             # a kernel this workload does not launch has no key
            res[kern] = {k: agg[k] / n[k] for k in sorted(agg)}
    return res

lres = pmc(("k_logistic_dense", "k_replay_condense", "k_collapse"), "lpmc")
lres["_note"] = ("mean per launch, summed over the device; separate --pmc passes of `bench.py --config regions5k --method logistic --steps 2 --warmup 1 "
                 "--no-cpu-baseline --no-extras` (24 regions of 5 kb, capture 120-250); FETCH_SIZE / WRITE_SIZE in KiB (FETCH_SIZE raw: double it for coalesced reads)")
json.dump(lres, open(f"{out}/{tag}_pmc_logistic.json", "w"), indent=1)
res = pmc(("k_svr_dense", "k_svr_finish", "k_records", "k_replay_condense"), "pmc")
res["_note"] = ("mean per launch, summed over the device; separate --pmc passes of `bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras` "
                "(default workload: practice62, capture 140-180, SVR n_sv=1024); FETCH_SIZE / WRITE_SIZE in KiB, raw")
json.dump(res, open(f"{out}/{tag}_pmc.json", "w"), indent=1)
k = res.get("k_svr_dense", {})
if "FETCH_SIZE" in k and "WRITE_SIZE" in k:
    fetch, write = k["FETCH_SIZE"] * 1024.0, k["WRITE_SIZE"] * 1024.0
    json.dump({"k_svr_dense_bytes_per_launch": 2.0 * fetch + write, "fetch_bytes_raw": fetch, "fetch_bytes_corrected": 2.0 * fetch, "write_bytes": write,
               "source": f"profiles/{tag}_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, tools/profile_round.sh; counter unit KiB)",
               "note": "FETCH_SIZE x 2: on gfx950 the counter tallies the 128-byte requests of coalesced reads at 64 bytes (MI355X_MICROARCH.md, HBM section); this "
                       "kernel reads 8-byte records and model rows with consecutive lanes on consecutive doubles, so the doubled value is the one to compare with a byte count"},
              open(f"{out}/{tag}_hbm_traffic.json", "w"), indent=1)
print(json.dumps(res.get("k_svr_dense", {}), indent=1))
d = summ["default"]
for kname in d:
    if any(t in kname for t in ("k_svr_dense", "k_records", "k_replay", "k_collapse")):
        print(kname[:60], json.dumps(d[kname]))
PY

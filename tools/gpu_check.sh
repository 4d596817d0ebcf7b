#!/bin/bash
# tools/gpu_check.sh TAG [quick] — GPU parity suite + bench lines of every workload config, on the MI355X box:
#   gpurun --timeout 2400 -- 'bash tools/gpu_check.sh r02a'
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd "$R"
timeout 1700 python3 -m pytest tests -m gpu -q --maxfail=6 --durations=12 > "$OUT/${TAG}_pytest.log" 2>&1
echo "pytest rc=$?"; tail -40 "$OUT/${TAG}_pytest.log"
python3 bench.py --steps 5 --warmup 2 > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"; echo "bench rc=$?"; tail -c 1500 "$OUT/${TAG}_bench.err"; cat "$OUT/${TAG}_bench.json"
python3 bench.py --config regions5k --regions 24 --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/${TAG}_bench_regions5k_logistic.json" 2>> "$OUT/${TAG}_bench.err"; cat "$OUT/${TAG}_bench_regions5k_logistic.json"
python3 bench.py --config regions5k --regions 24 --method svr --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/${TAG}_bench_regions5k_svr.json" 2>> "$OUT/${TAG}_bench.err"; cat "$OUT/${TAG}_bench_regions5k_svr.json"
python3 bench.py --config exome --regions 8192 --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/${TAG}_bench_exome.json" 2>> "$OUT/${TAG}_bench.err"; cat "$OUT/${TAG}_bench_exome.json"
python3 bench.py --config exome_snp --regions 4096 --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/${TAG}_bench_exome_snp.json" 2>> "$OUT/${TAG}_bench.err"; cat "$OUT/${TAG}_bench_exome_snp.json"
tail -c 1500 "$OUT/${TAG}_bench.err"

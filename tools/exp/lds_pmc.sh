cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "--config regions5k --method logistic" ""; do
D=/tmp/pl_$$; rm -rf $D
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $D -o p -- python3 $R/bench.py $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-parity-gate --no-measure-traffic > /dev/null 2>&1
python3 - $D <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for fn in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0][:50]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in agg:
    a = {c: agg[k][c] / n[k][c] for c in agg[k]}
    if a.get("SQ_INSTS_LDS", 0) > 1e4:
        print(f"{k:50s} insts {a['SQ_INSTS_LDS']:.3e} idx {a['SQ_LDS_IDX_ACTIVE']:.3e} confl {a['SQ_LDS_BANK_CONFLICT']:.3e}  non-conflict/instr {(a['SQ_LDS_IDX_ACTIVE']-a['SQ_LDS_BANK_CONFLICT'])/a['SQ_INSTS_LDS']:.2f}  conflict share {a['SQ_LDS_BANK_CONFLICT']/a['SQ_LDS_IDX_ACTIVE']:.2f}")
PY
done

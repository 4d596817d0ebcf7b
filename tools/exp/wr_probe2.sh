# WRITE_SIZE of k_svr_dense with its score stores compiled out, beside the product build (the excess write traffic on the
# small bench batch is a per-launch constant, not stores of the kernel).  Needs the scratch library first (in mipgen_amd/csrc, after `make`):
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSVR_NO_STORE -c kernels_svr.hip -o /tmp/kernels_svr_nostore.o
#   hipcc --offload-arch=gfx950 -shared -o ../../tools/exp/scratch/libmipgen_accel_nostore.so accel.o kernels_logistic.o /tmp/kernels_svr_nostore.o \
#         kernels_misc.o kernels_replay.o kernels_kmer.o kernels_logistic_dense.o kernels_format.o kernels_svr_gemm.o kernels_window.o kernels_skip.o
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for lib in tools/exp/scratch/libmipgen_accel_nostore.so ""; do
  rm -rf /tmp/wp
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/wp -o p -- python3 $R/tools/exp/kernel_time.py practice62 $lib > /dev/null 2>&1
  python3 - "$lib" <<'PY'
import csv, glob, sys
v=[]
for fn in glob.glob("/tmp/wp/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "k_svr_dense" in r["Kernel_Name"] and r["Counter_Name"]=="WRITE_SIZE": v.append(float(r["Counter_Value"])/1024)
print(sys.argv[1] or "product", "WRITE_SIZE MiB per launch:", [round(x,1) for x in v[:4]])
PY
done

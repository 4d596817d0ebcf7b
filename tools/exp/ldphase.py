# tools/exp/ldphase.py [regions5k|exome]: cycles per wavefront and stage of k_logistic_dense (a -DMIPGEN_DIAG scratch build: tools/exp/scratch/libmipgen_accel_diag.so)
import os, sys, ctypes as C, numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
from mipgen_amd import capi, workloads
capi.LIB_PATH = os.path.join(R, "tools", "exp", "scratch", "libmipgen_accel_diag.so")
lib = capi.load_library(capi.LIB_PATH); capi._lib = lib
cfg = sys.argv[1] if len(sys.argv) > 1 else "regions5k"
if cfg == "regions5k":
    P = capi.make_params(120, 250, score_method=capi.SCORE_LOGISTIC)
    acc = capi.Accel(P)
    regions = workloads.build_regions5k(acc, workloads.regions5k_genome(), workloads.regions5k_intervals(24), P, with_lrc=False)
else:
    chrom_len, all_iv = workloads.exome_layout()
    P = capi.make_params(150, 170, score_method=capi.SCORE_LOGISTIC)
    acc = capi.Accel(P)
    regions = workloads.build_exome(acc, chrom_len, all_iv[:8192], P)
acc.upload(regions)
acc.set_timing(True)
for _ in range(3):
    acc.score_window(0, capi.SCORE_LOGISTIC)
print(cfg, "k_logistic_dense ms", acc.last_kernel_ms(2))
buf = np.zeros(512 * 8 * 5, dtype=np.uint64)
rc = lib.mipgen_logistic_debug_dump(buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), buf.size)
b = buf.reshape(512, 8, 5).astype(np.float64)
nz = b.sum(axis=2) > 0
m = b[nz].mean(axis=0)
print("  ", dict(zip(["prologue + scans", "arm tables", "insert tables", "barrier", "rows"], np.round(m))), "cycles per wavefront of the first 512 workgroups; total", round(m.sum()), "shares", np.round(m / m.sum(), 3).tolist())

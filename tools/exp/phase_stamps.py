# tools/exp/phase_stamps.py CONFIG LIB: per-wavefront cycle stamps of k_svr_dense per SV group (scan / tables / candidate steps / barrier waits) from a -DMIPGEN_DIAG scratch build
import os, sys, ctypes as C, numpy as np, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
from mipgen_amd import capi, workloads
capi.LIB_PATH = os.path.join(R, sys.argv[2])
lib = capi.load_library(capi.LIB_PATH); capi._lib = lib
lib.mipgen_svr_debug_set.argtypes = [C.c_int]
cfg = sys.argv[1] if len(sys.argv) > 1 else "practice62"
if cfg == "practice62":
    genome, ivs = workloads.practice62()
    P = capi.make_params(140, 180, score_method=capi.SCORE_SVR)
    acc = capi.Accel(P)
    acc.load_model_file(workloads.svr_model_path("gpurun_out/bench_cache", genome, 1024))
    regions = workloads.build_regions(acc, genome, ivs, P)
else:
    chrom_len, all_iv = workloads.exome_layout()
    P = capi.make_params(150, 170, score_method=capi.SCORE_SVR)
    acc = capi.Accel(P)
    acc.load_model_file(workloads.svr_model_path("gpurun_out/bench_cache", workloads.practice62()[0], 1024))
    ivs = all_iv[:2048]
    if cfg.startswith("exomeK"):
        gr = acc.upload(workloads.build_exome(acc, chrom_len, all_iv[:4096], P))
        ivs = [iv for iv, g in zip(all_iv[:4096], gr) if g.n_sizes == int(cfg[6:])]
    regions = workloads.build_exome(acc, chrom_len, ivs, P)
acc.set_sv_split(1)
acc.upload(regions)
acc.set_timing(True)
n = acc.batch_candidates()
for dbg in (0,):
    lib.mipgen_svr_debug_set(dbg)
    ts = []
    for _ in range(3):
        acc.score_window(0, capi.SCORE_SVR); ts.append(acc.last_kernel_ms(0))
    print(f"dbg={dbg} (skip scan={dbg&1} tables={(dbg>>1)&1} cand={(dbg>>2)&1}): {min(ts):.3f} ms")
lib.mipgen_svr_debug_set(8)
acc.score_window(0, capi.SCORE_SVR); acc.last_kernel_ms(0)
buf = np.zeros(64 * 8 * 6 + 4096 * 8, dtype=np.uint64)
rc = lib.mipgen_svr_debug_dump(buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), buf.size)
b = buf[:64*8*6].reshape(32, 16, 6).astype(np.float64)
nz = b[:, :, 4] > 0
for w in range(16):
    m = b[:, w, :][b[:, w, 4] > 0]
    if len(m): print("wave", w, [int(round(v / 342.0)) for v in m.mean(axis=0)])
names = ["scan", "tables", "accumulate", "X-wait", "total", "Y-wait"]
it = 342.0
print({nm: round(v / it) for nm, v in zip(names, b[nz].mean(axis=0))}, "cycles per SV group per wave")
q = buf[64*8*6:].reshape(4096, 8); q = q[q[:, 3] > 0]
qq = q.astype(np.int64)
print('prologue us mean %.1f (min %.1f max %.1f); SV loop us mean %.1f; epilogue us mean %.1f (min %.1f max %.1f)' % (((qq[:,2]-qq[:,4])/100.0).mean(), ((qq[:,2]-qq[:,4])/100.0).min(), ((qq[:,2]-qq[:,4])/100.0).max(), ((qq[:,3]-qq[:,2])/100.0).mean(), ((qq[:,5]-qq[:,3])[qq[:,5]>qq[:,3]]/100.0).mean(), ((qq[:,5]-qq[:,3])[qq[:,5]>qq[:,3]]/100.0).min(), ((qq[:,5]-qq[:,3])[qq[:,5]>qq[:,3]]/100.0).max()), 'valid', int((qq[:,5]>qq[:,3]).sum()))
print("blocks", len(q), "mean block duration us", (q[:, 3].astype(np.int64) - q[:, 2].astype(np.int64)).mean() / 100.0, "kernel span us", (q[:,3].max()-q[:,2].min())/100.0)
# ---- block schedule analysis
import collections
t0 = q[:, 2].astype(np.int64); t1 = q[:, 3].astype(np.int64)
base = t0.min()
dur = (t1 - t0) / 100.0
print("block duration us: min %.0f p10 %.0f median %.0f p90 %.0f max %.0f; sum/256 = %.0f us; span %.0f us" % (dur.min(), np.percentile(dur, 10), np.median(dur), np.percentile(dur, 90), dur.max(), dur.sum() / 256, (t1.max() - base) / 100.0))

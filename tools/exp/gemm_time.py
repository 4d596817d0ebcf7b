# tools/exp/gemm_time.py [LIB] [N]: k_features_batch + k_svr_gemm on a list of N random candidates of practice62 (HIP-event kernel times), for A/B runs
import os, sys, hashlib, numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
from mipgen_amd import capi, workloads
lib_path = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] != "-" else None
N = int(sys.argv[2]) if len(sys.argv) > 2 else 250000
if lib_path:
    capi.LIB_PATH = os.path.join(R, lib_path); capi._lib = capi.load_library(capi.LIB_PATH)
genome, ivs = workloads.practice62()
P = capi.make_params(140, 180, score_method=capi.SCORE_SVR)
acc = capi.Accel(P)
acc.load_model_file(workloads.svr_model_path("gpurun_out/bench_cache", genome, 1024))
regions = workloads.build_regions(acc, genome, ivs, P)
acc.upload(regions)
acc.set_timing(True)
rng = np.random.default_rng(5)
pairs = capi.arm_pairs_of(P)
lc = []
for _ in range(N):
    ri = int(rng.integers(len(acc.grids))); g = acc.grids[ri]
    e, l = pairs[int(rng.integers(len(pairs)))]
    lc.append((ri, g.first_pos + int(rng.integers(g.n_pos)), P.max_capture_size - (g.first_size_index + int(rng.integers(g.n_sizes))) * P.capture_increment, e, l, int(rng.integers(2))))
ts = []
for _ in range(6):
    sc, rec, _, _ = acc.score_candidates(lc, capi.SCORE_SVR); ts.append((acc.last_kernel_ms(5), acc.last_kernel_ms(6)))
ts = np.array(ts[1:])
chk = hashlib.md5(np.ascontiguousarray(sc).tobytes()).hexdigest()[:12]
fl = 2.0 * N * 1024 * 192
print(f"gemm {lib_path or 'product'} N={N}: k_svr_gemm ms min {ts[:,0].min():.3f} ({fl / ts[:,0].min() / 1e9:.1f} TFLOP/s), k_features_batch ms {ts[:,1].min():.3f}, checksum {chk} sum {float(np.nansum(sc))!r}")

import os, sys, numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
from mipgen_amd import capi, workloads
lib_path = sys.argv[2] if len(sys.argv) > 2 else None
if lib_path:
    capi.LIB_PATH = os.path.join(R, lib_path); capi._lib = capi.load_library(capi.LIB_PATH)
cfg = sys.argv[1] if len(sys.argv) > 1 else "practice62"
if cfg.startswith("practice62"):
    genome, ivs = workloads.practice62()
    if ":" in cfg: ivs = ivs[:int(cfg.split(":")[1])]        # practice62:N = the first N regions, never cut along the SV list (traffic scans)
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
    if cfg.startswith("exomeK"):                # only the regions that keep K capture sizes (exomeK1 ... exomeK5)
        gr = acc.upload(workloads.build_exome(acc, chrom_len, all_iv[:4096], P))
        ivs = [iv for iv, g in zip(all_iv[:4096], gr) if g.n_sizes == int(cfg[6:])]
    regions = workloads.build_exome(acc, chrom_len, ivs, P)
if ":" in cfg: acc.set_sv_split(1)
grids = acc.upload(regions)
print(cfg, "regions", len(regions), "candidates", sum(g.count for g in grids), file=sys.stderr)
if len(sys.argv) > 3:                       # phase ablation of a -DMIPGEN_DIAG build: 1 no scan, 2 no tables, 4 no candidate steps (timing only)
    import ctypes
    capi._lib.mipgen_svr_debug_set(ctypes.c_int(int(sys.argv[3])))
acc.set_timing(True)
ts = []
for _ in range(12):
    acc.score_window(0, capi.SCORE_SVR); ts.append(acc.last_kernel_ms(0))
ts = np.array(ts[2:])
sc, rec = acc.download()
import hashlib
chk = hashlib.md5(np.ascontiguousarray(sc).tobytes()).hexdigest()[:12] + " sum=%r" % float(np.nansum(sc))
print(f"{cfg} {lib_path or 'product'} {' '.join(sys.argv[3:])}: kernel ms min {ts.min():.3f} median {np.median(ts):.3f}  checksum {chk!r}")

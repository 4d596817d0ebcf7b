# tools/exp/dump_scores.py CONFIG OUT.npy [LIB]: dense SVR scores of an A/B harness workload into a file (A/B comparisons of scratch builds: bit-identical?)
import os, sys, numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R)
from mipgen_amd import capi, workloads
if len(sys.argv) > 3:
    capi.LIB_PATH = os.path.join(R, sys.argv[3]); capi._lib = capi.load_library(capi.LIB_PATH)
cfg = sys.argv[1]
chrom_len, all_iv = workloads.exome_layout()
P = capi.make_params(150, 170, score_method=capi.SCORE_SVR)
acc = capi.Accel(P)
acc.load_model_file(workloads.svr_model_path("gpurun_out/bench_cache", workloads.practice62()[0], 1024))
n = int(cfg[5:]) if cfg.startswith("exome") and cfg[5:].isdigit() else 2048
grids = acc.upload(workloads.build_exome(acc, chrom_len, all_iv[:n], P))
acc.score_window(0, capi.SCORE_SVR)
sc, rec = acc.download()
np.save(sys.argv[2], sc)
np.save(sys.argv[2] + ".k.npy", np.repeat(np.array([g.n_sizes for g in grids]), np.array([g.count for g in grids])))

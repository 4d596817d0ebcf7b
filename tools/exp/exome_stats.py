import os, sys, numpy as np, collections
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R)
from mipgen_amd import capi, workloads
chrom_len, all_iv = workloads.exome_layout()
P = capi.make_params(150, 170, score_method=capi.SCORE_SVR)
acc = capi.Accel(P)
acc.load_model_file(workloads.svr_model_path("gpurun_out/bench_cache", workloads.practice62()[0], 1024))
regions = workloads.build_exome(acc, chrom_len, all_iv[:8192], P)
grids = acc.upload(regions)
c = collections.Counter(); cand = collections.Counter(); npos = collections.defaultdict(list)
for g in grids:
    c[g.n_sizes] += 1; cand[g.n_sizes] += g.count; npos[g.n_sizes].append(g.n_pos)
tot = sum(cand.values())
for k in sorted(c): print(f"K={k}: regions {c[k]}, candidates share {cand[k]/tot:.3f}, n_pos median {np.median(npos[k])} min {min(npos[k])} max {max(npos[k])}")

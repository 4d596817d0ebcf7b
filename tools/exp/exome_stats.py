# tools/exp/exome_stats.py: capture sizes per region of the synthetic exome, and what the regions of ONE size cost the dense SVR kernel per candidate
import os, sys, numpy as np, collections
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R)
from mipgen_amd import capi, workloads
chrom_len, all_iv = workloads.exome_layout()
P = capi.make_params(150, 170, score_method=capi.SCORE_SVR)
acc = capi.Accel(P)
acc.load_model_file(workloads.svr_model_path("gpurun_out/bench_cache", workloads.practice62()[0], 1024))
ivs = all_iv[:4096]
grids = acc.upload(workloads.build_exome(acc, chrom_len, ivs, P))
c = collections.Counter(); cand = collections.Counter(); npos = collections.defaultdict(list)
for g in grids:
    c[g.n_sizes] += 1; cand[g.n_sizes] += g.count; npos[g.n_sizes].append(g.n_pos)
tot = sum(cand.values())
for k in sorted(c): print(f"K={k}: regions {c[k]}, candidates share {cand[k]/tot:.3f}, n_pos median {np.median(npos[k])} min {min(npos[k])} max {max(npos[k])}")
acc.set_timing(True)
def timed(sel, name):
    sub = [iv for iv, g in zip(ivs, grids) if sel(g)]
    gr = acc.upload(workloads.build_exome(acc, chrom_len, sub, P))
    n = sum(g.count for g in gr)
    ts = []
    for _ in range(5):
        for w in range(acc.window_count()): acc.score_window(w, capi.SCORE_SVR)
        ts.append(acc.last_kernel_ms(0))
    print(f"{name}: {len(sub)} regions, {n} candidates, k_svr_dense {min(ts):.2f} ms = {n / min(ts) / 1e6:.3f} candidates/ns-ish (1e9/s: {n / min(ts) / 1e6:.3f})")
timed(lambda g: g.n_sizes == 1, "K = 1 only")
timed(lambda g: g.n_sizes == 5, "K = 5 only")
timed(lambda g: True, "all")

import os, sys, numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R)
from mipgen_amd import capi, workloads, synth
P = capi.make_params(120, 250, score_method=capi.SCORE_SVR)
genome = workloads.regions5k_genome()
ivs = workloads.regions5k_intervals(1)
cut = [synth.Interval(iv.chrom, iv.bed_start + 100, iv.bed_start + 100 + 260, iv.label) for iv in ivs]
acc = capi.Accel(P)
acc.load_model_file(workloads.svr_model_path("gpurun_out/bench_cache", workloads.practice62()[0], 1024, rho=-2.2))
acc.set_sv_split(1); acc.set_dynamic_skip(True)
regions = workloads.build_regions5k(acc, genome, cut, P)
grids = acc.upload(regions)
acc.score_window(0, capi.SCORE_SVR)
g = grids[0]
st, pb = acc.skip_state(g.n_pos)
print("n_pos", g.n_pos, "n_sizes", g.n_sizes, "state counts", np.bincount(st, minlength=3), "pb range", np.nanmin(pb), np.nanmax(pb), "skipped", acc.skipped_candidates())
s, r = acc.download()
A = P.n_arm_pairs
sc = s.reshape(g.n_pos, g.n_sizes, 2, A)
print("NaN rows per size index:", np.isnan(sc).all(axis=(2, 3)).sum(axis=0))
print("last pair max of size 0, list 0 (first 5 positions):", np.nanmax(sc[:5, 0, :, 11], axis=1))

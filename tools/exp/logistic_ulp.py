# tools/exp/logistic_ulp.py: the literal per-candidate logistic kernel (k_candidates) against the oracle (= the reference's double), in ulps
import os, sys, numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R)
from mipgen_amd import capi, workloads
from oracle import pyoracle as po
genome = workloads.regions5k_genome(); ivs = workloads.regions5k_intervals(1)
P = capi.make_params(120, 250, score_method=capi.SCORE_LOGISTIC)
regions = workloads.build_regions5k(None, genome, ivs, P, with_lrc=False)
acc = capi.Accel(P); grids = acc.upload(regions); g = grids[0]; A = P.n_arm_pairs
rng = np.random.default_rng(1); cands = []; refs = []
for idx in rng.choice(g.count, size=3000, replace=False):
    a = int(idx % A); row = int(idx // A); st = row & 1; rest = row >> 1; ki, pi = rest % g.n_sizes, rest // g.n_sizes
    c = (0, g.first_pos + pi, P.max_capture_size - (g.first_size_index + ki) * P.capture_increment, P.arm_ext[a], P.arm_lig[a], int(st))
    sk, d = po.design(P, regions[0], c)
    if sk: continue
    ref, _, _ = po.score_designed(d, capi.SCORE_LOGISTIC, np.zeros(44), None)
    if not np.isfinite(ref) or ref <= 0: continue
    cands.append(c); refs.append(ref)
got = acc.score_candidates(cands, capi.SCORE_LOGISTIC)[0]; refs = np.array(refs)
ulps = np.abs(got - refs) / np.spacing(refs)
print("n", len(refs), "exact", int((ulps == 0).sum()), "<=1", int((ulps <= 1).sum()), "<=4", int((ulps <= 4).sum()), "max", ulps.max(), "hist", np.histogram(ulps, bins=[0, 0.5, 1.5, 2.5, 4.5, 8.5, 16.5, 1e9])[0])

"""Hit rate of the exact dynamic skip between capture-size runs (mipgen.cpp:430): a scan position whose enumeration has ended inside the first run of
nine capture sizes constructs nothing in the later runs; a k_svr_dense tile of a later run could be skipped when ALL of its positions are in that state.
Measured from the replayed emitted masks of regions with more than nine capture sizes, for several rho of the synthetic model."""
import os, sys, numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R)
from mipgen_amd import capi, workloads
P = capi.make_params(120, 250, score_method=capi.SCORE_SVR)
genome = workloads.regions5k_genome()
ivs = workloads.regions5k_intervals(4)
A = P.n_arm_pairs
for rho in (-2.2, -2.25, -2.1, -1.645, -1.0):
    acc = capi.Accel(P)
    acc.load_model_file(workloads.svr_model_path("gpurun_out/bench_cache", workloads.practice62()[0], 1024, rho=rho))
    regions = workloads.build_regions5k(acc, genome, ivs, P)
    grids, scores, records = acc.score_regions(regions, capi.SCORE_SVR)
    acc.replay_condense()
    em, surv, mask = acc.download_replay()
    done1 = []; tiles = []
    for g in grids:
        m = mask[g.offset:g.offset + g.count].reshape(g.n_pos, g.n_sizes, 2 * A)
        later = m[:, 9:, :].any(axis=(1, 2))                 # anything constructed after the first run of nine sizes?
        done1.append(~later)
        d = ~later
        tiles += [d[i:i + 27].all() for i in range(0, g.n_pos, 27)]
    d = np.concatenate(done1)
    print(f"rho {rho}: emitted {em.sum() / sum(g.count for g in grids):.3f} of the dense grid; positions finished inside the first run {d.mean():.3f}; "
          f"27-position tiles of the later runs that could be skipped {np.mean(tiles):.3f}")
    acc.close()

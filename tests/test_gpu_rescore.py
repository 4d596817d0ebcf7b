"""GPU: mipgen_accel_rescore_survivors - the SVR score of every condensed survivor of a window, computed on the device from the survivor array
(mixed designs: /root/reference/mipgen.cpp:1523-1527, 1873-1877 re-score the tested MIPs one at a time) - equals mipgen_accel_score_candidates
on the same candidates, value for value, for long lists (features + matrix-core scorer) and short ones (the literal per-candidate kernel)."""
import numpy as np
import pytest

from mipgen_amd import capi
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _survivor_candidates(P, grids, surv):
    A = P.n_arm_pairs
    cands, slots = [], []
    pos = 0
    for ri, g in enumerate(grids):
        sv = surv[2 * pos:2 * (pos + g.n_pos)]
        for q in np.nonzero(sv["cand_index"] >= 0)[0]:
            idx = int(sv["cand_index"][q]) - g.offset
            a = idx % A; row = idx // A; st = row & 1; rest = row >> 1
            ki, pi = rest % g.n_sizes, rest // g.n_sizes
            cands.append((ri, g.first_pos + pi, P.max_capture_size - (g.first_size_index + ki) * P.capture_increment, P.arm_ext[a], P.arm_lig[a], int(st)))
            slots.append(2 * pos + int(q))
        pos += g.n_pos
    return cands, np.array(slots, dtype=np.int64)


@pytest.mark.parametrize("name,n_regions", [("mixed_12_regions", None), ("mixed_small", 1)])
def test_rescore_survivors_equals_the_list_scorer(name, n_regions):
    meta = H.load_design(name)
    genome = H.golden_genome(meta.get("genome", "genome_chr1.fa.gz"))
    P = H.design_params(meta, capi.SCORE_LOGISTIC)
    acc = capi.Accel(P)
    import os
    acc.load_model_file(os.path.join(H.GOLDEN, "models", meta["model"]))
    regions = H.design_regions(meta, genome, P, lrc_fn=lambda s, a, b: acc.long_range_content(s, a, b))
    if n_regions:
        regions = regions[:n_regions]
    grids = acc.upload(regions)
    acc.score_window(0, capi.SCORE_LOGISTIC)
    acc.replay_condense()
    _, surv, _ = acc.download_replay()
    svr = acc.rescore_survivors(0)
    cands, slots = _survivor_candidates(P, grids, surv)
    assert len(cands) > 0 and svr.shape[0] == surv.shape[0]
    ref = acc.score_candidates(cands, capi.SCORE_SVR)[0]
    assert np.array_equal(svr[slots], ref, equal_nan=True)                       # the same kernels on the same list: value for value
    empty = np.ones(svr.shape[0], dtype=bool); empty[slots] = False
    assert np.all(np.isnan(svr[empty]))
    # a short list takes the literal per-candidate kernel in both routes
    if len(cands) >= 256:
        few = capi.Accel(P)
        few.load_model_file(os.path.join(H.GOLDEN, "models", meta["model"]))
        g1 = few.upload(regions[:1])
        few.score_window(0, capi.SCORE_LOGISTIC); few.replay_condense()
        _, s1, _ = few.download_replay()
        c1, sl1 = _survivor_candidates(P, g1, s1)
        if 0 < len(c1) < 256:
            v1 = few.rescore_survivors(0)
            assert np.array_equal(v1[sl1], few.score_candidates(c1, capi.SCORE_SVR)[0], equal_nan=True)
        few.close()
    acc.close()

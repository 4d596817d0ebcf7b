"""GPU: the reference's dynamic skip between capture sizes (mipgen.cpp:430) applied between the capture-size RUNS of the dense SVR scorer
(mipgen_accel_set_dynamic_skip; kernels_skip.hip).  A tile of a later run is left out when every one of its positions has stopped constructing
candidates before it.  Whatever the model does - every position stops inside the first run, inside the second, or none at all - the replayed
emitted mask, the emitted counts, the condensed survivors and the collapse result must be IDENTICAL to the ones of the full dense grid, the dense
scores must be identical wherever a candidate is emitted, and the replay + condense of the oracle over the full grid must agree."""
import numpy as np
import pytest

from mipgen_amd import capi, workloads
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
CACHE = "/tmp/mipgen_test_cache"


def _batch(acc, P):
    genome = workloads.regions5k_genome()
    ivs = [iv for iv in workloads.regions5k_intervals(3)]
    # 3 x 5 kb would be 48 M candidates: cut the intervals down to 260 / 90 / 400 bases (27, 1..n and 27 capture sizes after the static skip)
    from mipgen_amd import synth
    cut = [synth.Interval(iv.chrom, iv.bed_start + 100, iv.bed_start + 100 + n, iv.label) for iv, n in zip(ivs, (260, 90, 400))]
    return workloads.build_regions5k(acc, genome, cut, P)


@pytest.mark.parametrize("rho,expect", [(-2.3, "first_run"), (-2.1, "later"), (-1.0, "never")])
def test_dynamic_skip_changes_nothing_but_the_work(rho, expect):
    P = capi.make_params(120, 250, score_method=capi.SCORE_SVR)
    mp = workloads.svr_model_path(CACHE, workloads.practice62()[0], 1024, rho=rho)
    out = {}
    for on in (False, True):
        acc = capi.Accel(P)
        acc.load_model_file(mp)
        acc.set_sv_split(1)                                   # the skipping launch never splits along the SV list: same summation order in both runs
        acc.set_dynamic_skip(on)
        regions = _batch(acc, P)
        grids = acc.upload(regions)
        assert max(g.n_sizes for g in grids) == 27
        acc.score_window(0, capi.SCORE_SVR)
        scores, records = acc.download()
        acc.replay_condense()
        em, surv, mask = acc.download_replay()
        acc.collapse()
        col = acc.download_collapsed()
        skipped = acc.skipped_candidates()
        # the silent path takes the same route
        acc.score_condense_all(capi.SCORE_SVR)
        em2, surv2 = acc.download_survivors()
        out[on] = dict(scores=scores, records=records, em=em.copy(), surv=surv.copy(), mask=mask.copy(), col=col.copy(), skipped=skipped, em2=em2.copy(),
                       surv2=surv2.copy(), grids=grids, regions=regions)
        acc.close()
    a, b = out[False], out[True]
    assert a["skipped"] == 0
    assert np.array_equal(a["records"], b["records"])
    assert np.array_equal(a["mask"], b["mask"]) and np.array_equal(a["em"], b["em"])
    assert a["surv"].tobytes() == b["surv"].tobytes() and np.array_equal(a["col"], b["col"])
    assert np.array_equal(a["em2"], b["em2"])
    assert np.array_equal(a["surv2"]["cand_index"], b["surv2"]["cand_index"]) and np.array_equal(a["surv2"]["record"], b["surv2"]["record"])
    assert np.array_equal(a["surv2"]["score"], b["surv2"]["score"], equal_nan=True)
    emitted = a["mask"] != 0
    assert np.array_equal(a["scores"][emitted], b["scores"][emitted], equal_nan=True)       # every constructed candidate carries the same score
    left_out = np.isnan(b["scores"]) & ~np.isnan(a["scores"])
    assert not (left_out & emitted).any() and int(left_out.sum()) == b["skipped"]
    dense = sum(g.count for g in a["grids"])
    # what the full grid's own emitted mask says about the positions of the 27-size regions: stopped inside the first run of nine sizes?
    A = P.n_arm_pairs
    done1 = np.concatenate([~a["mask"][g.offset:g.offset + g.count].reshape(g.n_pos, g.n_sizes, 2 * A)[:, 9:, :].any(axis=(1, 2))
                            for g in a["grids"] if g.n_sizes == 27])
    print(f"rho {rho}: emitted {a['em'].sum() / dense:.3f} of the grid, positions stopped inside the first run {done1.mean():.3f}, skipped {b['skipped'] / dense:.3f}")
    if expect == "first_run":
        assert done1.mean() > 0.9 and b["skipped"] > 0.4 * dense      # (nearly) two of the three runs of the 27-size regions
    elif expect == "never":
        assert done1.mean() == 0 and b["skipped"] == 0
    # the oracle's replay + condense over the FULL grid agrees with the survivors of the skipping run
    pos0 = 0
    for rd, g in zip(a["regions"], a["grids"]):
        s = a["scores"][g.offset:g.offset + g.count]
        r = a["records"][g.offset:g.offset + g.count]
        n_emit, omask = po.replay_region(P, rd, s, r)
        osurv = po.condense_region(P, rd, s, r, omask)
        mine = b["surv"][2 * pos0:2 * (pos0 + g.n_pos)]
        assert int(b["em"][a["grids"].index(g)]) == n_emit
        assert np.array_equal(np.where(mine["cand_index"] >= 0, mine["cand_index"] - g.offset, -1), osurv["cand_index"])
        pos0 += g.n_pos


def _fuzz_configs():
    import os
    rng = np.random.default_rng(20261003)
    grid = [(e, l) for e in range(16, 31) for l in range(18, 31) if 38 <= e + l <= 47]
    out = []
    for i in range(int(os.environ.get("MIPGEN_FUZZ_SKIP_N", "8"))):
        inc = int(rng.choice([2, 3, 5, 7]))
        K = int(rng.integers(10, 31))
        lo = int(rng.integers(100, 140))
        n_pairs = int(rng.integers(2, 58))
        idx = sorted(rng.permutation(len(grid))[:n_pairs].tolist(), key=lambda j: (-(grid[j][0] + grid[j][1]), grid[j][0]))
        out.append((i, lo, lo + inc * (K - 1), inc, [grid[j] for j in idx], float(rng.uniform(-3.2, -1.2)), float(rng.uniform(1.6, 2.8)),
                    int(rng.integers(5000, 9_000_000)), int(rng.choice([120, 250, 600]))))
    return out


@pytest.mark.parametrize("cfg", _fuzz_configs(), ids=lambda c: f"skip{c[0]}_C{c[1]}-{c[2]}x{c[3]}_A{len(c[4])}_rho{c[5]:.2f}_up{c[6]:.2f}")
def test_dynamic_skip_random_configurations(cfg):
    """Random capture ranges of 10..30 sizes (2..4 runs), arm-pair subsets, rho and upper score limit: the skipping launch and the full one give the
    same emitted mask, emitted counts and survivors, and never leave out a constructed candidate."""
    i, lo, hi, inc, pairs, rho, upper, start, length = cfg
    from mipgen_amd import synth
    P = capi.make_params(lo, hi, score_method=capi.SCORE_SVR, capture_increment=inc, arm_pairs=pairs, svr_optimal=upper)
    mp = workloads.svr_model_path(CACHE, workloads.practice62()[0], 256, rho=rho)
    genome = workloads.regions5k_genome()
    iv = [synth.Interval("1", start, start + length, f"z{i}")]
    res = {}
    for on in (False, True):
        acc = capi.Accel(P)
        acc.load_model_file(mp)
        acc.set_sv_split(1)
        acc.set_dynamic_skip(on)
        regions = workloads.build_regions5k(acc, genome, iv, P)
        grids = acc.upload(regions)
        acc.score_window(0, capi.SCORE_SVR)
        scores, records = acc.download()
        acc.replay_condense()
        em, surv, mask = acc.download_replay()
        res[on] = (scores, records, em.copy(), surv.copy(), mask.copy(), acc.skipped_candidates(), grids[0].n_sizes)
        acc.close()
    a, b = res[False], res[True]
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and np.array_equal(a[4], b[4])
    assert a[3].tobytes() == b[3].tobytes()
    emitted = a[4] != 0
    assert np.array_equal(a[0][emitted], b[0][emitted], equal_nan=True)
    left_out = np.isnan(b[0]) & ~np.isnan(a[0])
    assert not (left_out & emitted).any() and int(left_out.sum()) == b[5]
    print(f"SKIPSTAT sizes {b[6]} emitted {emitted.mean():.3f} skipped {b[5] / max(1, a[0].size):.3f}")


def test_dynamic_skip_against_the_automatic_sv_split():
    """What the command line did before the skip against what it does with it: one launch per window cut along the SV list by the library's own policy
    (partial sums added in part order) versus one launch per capture-size run without a split.  The summation orders differ, so scores may differ in
    their last bits - nothing else may: the same candidates are constructed, the same survivors chosen, the same records printed."""
    P = capi.make_params(120, 250, score_method=capi.SCORE_SVR)
    mp = workloads.svr_model_path(CACHE, workloads.practice62()[0], 1024, rho=-2.2)
    out = {}
    for on in (False, True):
        acc = capi.Accel(P)
        acc.load_model_file(mp)
        acc.set_dynamic_skip(on)                              # (no set_sv_split: the automatic policy - this small batch is cut along the SV list when the skip is off)
        regions = _batch(acc, P)
        acc.upload(regions)
        acc.score_window(0, capi.SCORE_SVR)
        scores, records = acc.download()
        acc.replay_condense()
        em, surv, mask = acc.download_replay()
        out[on] = dict(scores=scores, records=records, em=em.copy(), surv=surv.copy(), mask=mask.copy())
        acc.close()
    a, b = out[False], out[True]
    assert np.array_equal(a["records"], b["records"]) and np.array_equal(a["mask"], b["mask"]) and np.array_equal(a["em"], b["em"])
    assert np.array_equal(a["surv"]["cand_index"], b["surv"]["cand_index"]) and np.array_equal(a["surv"]["record"], b["surv"]["record"])
    emitted = a["mask"] != 0
    d = np.abs(a["scores"][emitted] - b["scores"][emitted])
    assert np.nanmax(d) < 1e-11                               # another summation order, nothing more
    fmt = np.frompyfunc(lambda x: format(x, ".6g"), 1, 1)
    have = a["surv"]["cand_index"] >= 0
    assert np.array_equal(fmt(a["surv"]["score"][have]), fmt(b["surv"]["score"][have]))      # what a front end would print for the survivors

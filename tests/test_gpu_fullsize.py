"""GPU: properties at BASELINE.json's full config-2 size (practice62, capture 140-180, 57 arm pairs, SVR with a 1024-SV model:
15.5 M dense candidates), where the oracle is too slow to score everything:
  * additivity of the SVR decision function over disjoint support-vector subsets (linearity in the model),
  * reverse-complement symmetry: a plus-strand candidate and its mirror image on the reverse-complemented genome
    (scored through the minus-strand code path) have the same oriented sequences, hence the same scores and counts,
  * the window-separable dense kernel agrees with the direct 192-dimension sparse kernel on a random sample, and a
    sub-sample agrees with the oracle within 1e-5,
  * bitwise determinism, and the invariants of replay + condense.
"""
import os

import numpy as np
import pytest

from mipgen_amd import capi, synth, workloads
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
CACHE = "/tmp/mipgen_test_cache"


@pytest.fixture(scope="module")
def full():
    genome, ivs = workloads.practice62()
    P = capi.make_params(140, 180, score_method=capi.SCORE_SVR)
    acc = capi.Accel(P)
    mp = workloads.svr_model_path(CACHE, genome, 1024)
    acc.load_model_file(mp)
    regions = workloads.build_regions(acc, genome, ivs, P)
    grids = acc.upload(regions)
    acc.score_resident(capi.SCORE_SVR)
    scores, records = acc.download()
    yield dict(genome=genome, ivs=ivs, P=P, acc=acc, model=mp, regions=regions, grids=grids, scores=scores, records=records)
    acc.close()


def _decode(P, g, idx):
    A = P.n_arm_pairs
    a = idx % A
    row = idx // A
    strand = row & 1
    rest = row >> 1
    ki, pi = rest % g.n_sizes, rest // g.n_sizes
    return pi, ki, a, strand


def test_full_size_determinism_and_counts(full):
    acc = full["acc"]
    assert acc.batch_candidates() == full["scores"].size > 15_000_000
    acc.score_resident(capi.SCORE_SVR)
    s2, r2 = acc.download()
    assert np.array_equal(s2.view(np.uint64), full["scores"].view(np.uint64))         # bitwise
    assert np.array_equal(r2, full["records"])
    valid = (capi.rec_flags(full["records"]) & capi.FLAG_VALID) != 0
    assert valid.mean() > 0.99
    assert np.isfinite(full["scores"][valid]).all()


def test_sv_subset_additivity(full):
    """score + rho is a sum over support vectors: (all) == (first half) + (second half)."""
    om = po.Model(full["model"])
    sv, coef = om.densify()
    acc = full["acc"]
    h = sv.shape[0] // 2
    parts = []
    for lo, hi in ((0, h), (h, sv.shape[0])):
        acc.set_model(om.gamma, 0.0, coef[lo:hi], sv[lo:hi])
        acc.score_resident(capi.SCORE_SVR)
        s, _ = acc.download()
        parts.append(s)
    acc.load_model_file(full["model"])
    fl = capi.rec_flags(full["records"])
    normal = ((fl & capi.FLAG_VALID) != 0) & ((fl & capi.FLAG_GUARD) == 0) & (capi.rec_ext_copy(full["records"]) > 0) & (capi.rec_lig_copy(full["records"]) > 0)
    lhs = full["scores"][normal] + om.rho
    rhs = parts[0][normal] + parts[1][normal]
    assert np.max(np.abs(lhs - rhs)) < 1e-9


def test_dense_vs_sparse_kernel_and_oracle_sample(full):
    rng = np.random.default_rng(11)
    acc, P = full["acc"], full["P"]
    om = po.Model(full["model"])
    worst = 0.0
    n_oracle = 0
    for ri in rng.choice(len(full["regions"]), size=6, replace=False):
        g, rd = full["grids"][ri], full["regions"][ri]
        rec = full["records"][g.offset:g.offset + g.count]
        valid = np.nonzero((capi.rec_flags(rec) & capi.FLAG_VALID) != 0)[0]
        pick = rng.choice(valid, size=400, replace=False)
        cands = []
        for idx in pick:
            pi, ki, a, strand = _decode(P, g, int(idx))
            cands.append((int(ri), g.first_pos + pi, P.max_capture_size - (g.first_size_index + ki) * P.capture_increment,
                          P.arm_ext[a], P.arm_lig[a], int(strand)))
        s_sparse, r_sparse, _, _ = acc.score_candidates(cands, capi.SCORE_SVR)
        s_dense = full["scores"][g.offset + pick]
        assert np.array_equal(r_sparse, rec[pick])
        worst = max(worst, float(np.max(np.abs(s_sparse - s_dense))))
        for j in range(12):                                            # oracle: slow, a few per region
            c = cands[j]
            sk, d = po.design(P, rd, (0,) + c[1:])
            so, _, _ = po.score_designed(d, capi.SCORE_SVR, np.array(rd.c.long_range_content[:]), om)
            assert abs(so - s_dense[j]) <= 1e-5, (c, so, s_dense[j])
            n_oracle += 1
    assert worst < 1e-8, worst                                         # two GPU formulations of the same sum
    assert n_oracle == 72


@pytest.mark.parametrize("method", [capi.SCORE_LOGISTIC, capi.SCORE_SVR])
def test_reverse_complement_symmetry(full, method):
    """(p, C, e, l, '+') on the forward genome  <->  (G + 2 - p - ss, C, e, l, '-') on its reverse complement."""
    genome, P, acc = full["genome"], full["P"], full["acc"]
    G = len(genome)
    rc = synth.revcomp(genome.decode()).encode()
    lrc = np.linspace(0.02, 0.25, 44)
    worst = 0.0
    n_checked = 0
    for iv in full["ivs"][:8]:
        fwd = capi.build_region(genome, "7", iv.bed_start, iv.bed_end, P, lrc=lrc)
        s, t = iv.bed_start + 1, iv.bed_end
        mir = capi.build_region(rc, "7", G - t, G + 1 - s, P, lrc=lrc)
        grids, scores, records = acc.score_regions([fwd, mir], method)
        gf, gm = grids
        A = P.n_arm_pairs
        idx = np.arange(gf.count, dtype=np.int64)
        a = idx % A
        row = idx // A
        strand = row & 1
        rest = row >> 1
        ki, pi = rest % gf.n_sizes, rest // gf.n_sizes
        e = np.array([P.arm_ext[i] for i in range(A)])[a]
        l = np.array([P.arm_lig[i] for i in range(A)])[a]
        C = P.max_capture_size - (gf.first_size_index + ki) * P.capture_increment
        ss = C - e - l
        p = gf.first_pos + pi
        pm = G + 2 - p - ss
        pim = pm - gm.first_pos
        ok = (strand == 0) & (pim >= 0) & (pim < gm.n_pos)
        midx = (((pim * gm.n_sizes + ki) * 2 + 1) * A + a)[ok]
        fidx = idx[ok]
        rf, rm = records[gf.offset + fidx], records[gm.offset + midx]
        both = ((capi.rec_flags(rf) & capi.FLAG_VALID) != 0) & ((capi.rec_flags(rm) & capi.FLAG_VALID) != 0)
        assert both.sum() > 1000
        assert np.array_equal(capi.rec_junction(rf[both]), capi.rec_junction(rm[both]))      # same oriented ligation arm
        d = np.abs(scores[gf.offset + fidx][both] - scores[gm.offset + midx][both])
        worst = max(worst, float(np.nanmax(d)))
        n_checked += int(both.sum())
    assert worst < 1e-9, worst
    assert n_checked > 100_000
    # restore the resident batch of the module fixture
    acc.upload(full["regions"])
    acc.score_resident(capi.SCORE_SVR)


def test_replay_condense_invariants_full_size(full):
    acc = full["acc"]
    acc.upload(full["regions"])
    acc.score_resident(capi.SCORE_SVR)
    acc.replay_condense()
    emitted, surv, mask = acc.download_replay()
    valid = (capi.rec_flags(full["records"]) & capi.FLAG_VALID) != 0
    assert not (mask.astype(bool) & ~valid).any()                     # only constructible candidates are emitted
    assert int(mask.sum()) == int(emitted.sum())
    # emission is by (plus, minus) pairs: rows of A plus-strand flags are followed by the same A minus-strand flags
    A = full["P"].n_arm_pairs
    m2 = mask.reshape(-1, 2, A)
    assert np.array_equal(m2[:, 0, :], m2[:, 1, :])
    have = surv["cand_index"] >= 0
    assert mask[surv["cand_index"][have]].all()                       # survivors are emitted candidates
    assert np.array_equal(surv["score"][have], full["scores"][surv["cand_index"][have]])
    assert (((surv["cand_index"][have] // A) % 2) == (np.nonzero(have)[0] % 2)).all()     # survivor slot parity = strand

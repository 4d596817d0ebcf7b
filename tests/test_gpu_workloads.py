"""GPU: the BASELINE.json configs beyond the exon-sized practice design - 5 kb regions (configs[2]), exome-shaped batches
(configs[3]) and the tags + SNP variant (configs[4]) - against the oracle.  Long regions reach code the short ones never do: position
tiling over thousands of scan starts, all 27 capture sizes on every tile, N runs that trip the guard and the `skip_ahead` exit
(/root/reference/mipgen.cpp:494-497), batches cut into result windows.
"""
import os

import numpy as np
import pytest

from mipgen_amd import capi, synth, workloads
from oracle import pyoracle as po
from tests import helpers as H

pytestmark = pytest.mark.gpu
TOL = 1e-5
CACHE = "/tmp/mipgen_test_cache"


def _close(a, b, tol=TOL):
    a, b = np.asarray(a), np.asarray(b)
    nan_ok = np.isnan(a) & np.isnan(b)
    inf_ok = np.isinf(a) & (a == b)
    with np.errstate(invalid="ignore"):
        d = np.abs(a - b)
    return (nan_ok | inf_ok | (d <= tol)), (np.nanmax(np.where(nan_ok | inf_ok, 0.0, d)) if a.size else 0.0)


@pytest.fixture(scope="module")
def g5k():
    return workloads.regions5k_genome()


def _decode(P, g, idx):
    A = P.n_arm_pairs
    a = idx % A
    row = idx // A
    strand = row & 1
    rest = row >> 1
    ki, pi = rest % g.n_sizes, rest // g.n_sizes
    return (0, g.first_pos + int(pi), P.max_capture_size - (g.first_size_index + int(ki)) * P.capture_increment,
            P.arm_ext[int(a)], P.arm_lig[int(a)], int(strand))


def test_regions5k_logistic_full_grid(g5k):
    """configs[2] shape: one 5,000-bp region holding a run of 50 N, capture 120-250 (27 sizes x 57 pairs x 2 strands at each of
    5,205 scan starts = 16 M candidates): every record bit-exact, every logistic score within 1e-5, the -1000 guard exact; replay +
    condense identical to the oracle's, with the heuristic exit of mipgen.cpp:494 firing."""
    ivs = workloads.regions5k_intervals(1, first=1)
    assert g5k[ivs[0].bed_start:ivs[0].bed_end].count(b"N") >= 50
    P = capi.make_params(120, 250, score_method=capi.SCORE_LOGISTIC)
    rd = workloads.build_regions5k(None, g5k, ivs, P, with_lrc=False)[0]
    acc = capi.Accel(P)
    acc.set_logistic_subruns(4)                          # as the large batches of this shape run: four position sub-runs per workgroup, sliding tables
    grids, scores, records = acc.score_regions([rd], capi.SCORE_LOGISTIC)
    g = grids[0]
    assert g.n_sizes == 27 and g.count > 15_000_000
    for subs in (1, 3):                                  # the layout does not change a bit of the results
        acc.set_logistic_subruns(subs)
        acc.score_window(0, capi.SCORE_LOGISTIC)
        s2, r2 = acc.download()
        assert np.array_equal(r2, records) and np.array_equal(s2, scores, equal_nan=True), subs
    acc.set_logistic_subruns(4)
    acc.score_window(0, capi.SCORE_LOGISTIC)
    og, os_, or_ = po.score_region_dense(P, rd, capi.SCORE_LOGISTIC, None)
    assert og.count == g.count
    bad = np.nonzero(records != or_)[0]
    assert bad.size == 0, (int(bad[0]), hex(int(records[bad[0]])), hex(int(or_[bad[0]])))
    ok, mx = _close(scores, os_)
    assert ok.all(), (mx, int(np.nonzero(~ok)[0][0]))
    flags = capi.rec_flags(records)
    guard = ((flags & capi.FLAG_GUARD) != 0) & ((flags & capi.FLAG_VALID) != 0)
    assert guard.sum() > 10_000 and np.all(scores[guard] == -1000.0)
    assert (capi.rec_ext_copy(records) == 101).any() and (capi.rec_lig_copy(records) > 1).any()
    acc.replay_condense()
    emitted, surv, mask = acc.download_replay()
    n_emit, omask = po.replay_region(P, rd, scores, records)
    assert emitted[0] == n_emit and np.array_equal(mask, omask)
    assert n_emit < int(((flags & capi.FLAG_VALID) != 0).sum())           # the early exits removed candidates
    osurv = po.condense_region(P, rd, scores, records, omask)
    assert np.array_equal(surv["cand_index"], osurv["cand_index"])
    assert np.array_equal(surv["score"], osurv["score"], equal_nan=True)
    assert np.array_equal(surv["record"], osurv["record"])
    acc.close()


def test_regions5k_svr_sampled_and_windows(g5k):
    """Dense SVR over 5 kb regions (three capture-size runs per position tile at K = 27), two regions forced into two result windows:
    2,000 random candidates + every guard / zero-copy special against the oracle; survivors of score_condense_all equal the
    window-by-window replay."""
    ivs = workloads.regions5k_intervals(2, first=5)
    P = capi.make_params(120, 250, score_method=capi.SCORE_SVR)
    mp = workloads.svr_model_path(CACHE, workloads.practice62()[0], 96, seed=5)
    acc = capi.Accel(P)
    acc.load_model_file(mp)
    om = po.Model(mp)
    regions = workloads.build_regions5k(acc, g5k, ivs, P)
    for rd in regions:                                                          # device long-range content == oracle's
        n = rd.c.seq_stop - rd.c.seq_start + 1
        s0 = rd.c.start_flanked - P.max_capture_size - 1 - 1000
        assert np.array_equal(np.array(rd.c.long_range_content[:]), po.long_range_content(g5k[s0:s0 + n + 2000], rd.c.seq_start, rd.c.seq_stop))
    acc.set_window_candidates(17_000_000)
    acc.upload(regions)
    assert acc.window_count() == 2
    rng = np.random.default_rng(12)
    all_surv = []
    for w in range(2):
        wi = acc.window_info(w)
        acc.score_window(w, capi.SCORE_SVR)
        scores, records = acc.download(wi["first_candidate"], wi["n_candidates"])
        rd, g = regions[w], acc.grids[w]
        flags = capi.rec_flags(records)
        valid = np.nonzero((flags & capi.FLAG_VALID) != 0)[0]
        special = valid[((flags[valid] & capi.FLAG_GUARD) != 0) | (capi.rec_ext_copy(records[valid]) == 0) | (capi.rec_lig_copy(records[valid]) == 0)]
        pick = np.unique(np.concatenate([rng.choice(valid, size=1000, replace=False), rng.choice(special, size=min(100, special.size), replace=False)]))
        lrc = np.array(rd.c.long_range_content[:])
        for idx in pick:
            cand = _decode(P, g, int(idx))
            sk, d = po.design(P, rd, cand)
            assert not sk
            so, _, _ = po.score_designed(d, capi.SCORE_SVR, lrc, om)
            ok, _ = _close([scores[idx]], [so])
            assert ok.all(), (w, cand, scores[idx], so)
        acc.replay_condense()
        e, v, m = acc.download_replay(window=w)
        n_emit, omask = po.replay_region(P, rd, scores, records)
        assert e[0] == n_emit and np.array_equal(m, omask)
        osurv = po.condense_region(P, rd, scores, records, omask)
        assert np.array_equal(v["cand_index"], np.where(osurv["cand_index"] >= 0, osurv["cand_index"] + g.offset, -1))
        all_surv.append(v)
    acc.score_condense_all(capi.SCORE_SVR)
    e_all, v_all = acc.download_survivors()
    ref = np.concatenate(all_surv)
    for f in ("cand_index", "record"):
        assert np.array_equal(v_all[f], ref[f]), f
    assert np.array_equal(v_all["score"], ref["score"], equal_nan=True)
    acc.close()


def test_exome_shard_with_snps():
    """configs[3]/[4] shape: 400 exon-like regions of the synthetic exome (ragged lengths 20 .. several kb: K = 1 .. 27 surviving capture
    sizes) with the SNP classes of 1 SNP / 300 bp, capture 120-250.  Records (incl. SNP counts / flags) bit-exact and logistic scores
    within 1e-5 on every candidate of 12 regions spread over the length range; replay + condense identical on those; SVR on a sample."""
    chrom_len, ivs = workloads.exome_layout()
    P = capi.make_params(120, 250, score_method=capi.SCORE_SVR)
    shard = ivs[1000:1400]
    mp = workloads.svr_model_path(CACHE, workloads.practice62()[0], 64, seed=9)
    acc = capi.Accel(P)
    acc.load_model_file(mp)
    om = po.Model(mp)
    regions = workloads.build_exome(acc, chrom_len, shard, P, snps=True)
    grids = acc.upload(regions)
    assert len({g.n_sizes for g in grids}) >= 5                            # ragged capture-size sets
    w = workloads.dense_candidates(shard, P)
    assert np.array_equal(w, np.array([g.count for g in grids]))            # the shard weights are the true grid sizes
    order = np.argsort([iv.bed_end - iv.bed_start for iv in shard])
    chosen = [int(order[i]) for i in np.linspace(0, len(order) - 1, 12).astype(int)]
    acc.score_window(0, capi.SCORE_LOGISTIC)
    scores, records = acc.download()
    acc.replay_condense()
    emitted, surv, mask = acc.download_replay()
    pos_off = np.concatenate([[0], np.cumsum([g.n_pos for g in grids])])
    n_snp = 0
    for ri in chosen:
        rd, g = regions[ri], grids[ri]
        _, os_, or_ = po.score_region_dense(P, rd, capi.SCORE_LOGISTIC, None)
        s, r = scores[g.offset:g.offset + g.count], records[g.offset:g.offset + g.count]
        assert np.array_equal(r, or_), ri
        ok, mx = _close(s, os_)
        assert ok.all(), (ri, mx)
        n_snp += int((capi.rec_snp_count(r) > 0).sum())
        n_emit, omask = po.replay_region(P, rd, s, r)
        assert emitted[ri] == n_emit and np.array_equal(mask[g.offset:g.offset + g.count], omask)
        osurv = po.condense_region(P, rd, s, r, omask)
        got = surv[2 * pos_off[ri]:2 * pos_off[ri + 1]]
        assert np.array_equal(got["cand_index"], np.where(osurv["cand_index"] >= 0, osurv["cand_index"] + g.offset, -1))
        assert np.array_equal(got["record"], osurv["record"])
    assert n_snp > 1000
    acc.score_window(0, capi.SCORE_SVR)
    scores, records = acc.download()
    rng = np.random.default_rng(2)
    for ri in chosen[::3]:
        rd, g = regions[ri], grids[ri]
        r = records[g.offset:g.offset + g.count]
        valid = np.nonzero((capi.rec_flags(r) & capi.FLAG_VALID) != 0)[0]
        lrc = np.array(rd.c.long_range_content[:])
        for idx in rng.choice(valid, size=min(250, valid.size), replace=False):
            cand = _decode(P, g, int(idx))
            sk, d = po.design(P, rd, cand)
            so, _, _ = po.score_designed(d, capi.SCORE_SVR, lrc, om)
            ok, _ = _close([scores[g.offset + idx]], [so])
            assert ok.all(), (ri, cand, scores[g.offset + idx], so)
    acc.close()


@pytest.mark.parametrize("n_sv,gamma,coef_scale", [(256, None, 1.0), (4096, None, 1.0), (300, 0.05 / 192.0 * 40.0 * 20.0, 50.0)])
def test_model_sweep(n_sv, gamma, coef_scale):
    """The nSV sweep of SURVEY.md section 8d (256 / 4096; 1024 is the bench model) and a model with gamma x 20 and |coef| up to 50:
    the dense kernel keeps every window norm and every exponential in full double precision, so the score error stays ~1e-12 *
    sum|coef| for any model.  Sampled against the oracle, and against the direct 192-dimension kernel on the device."""
    genome, ivs = workloads.practice62()
    mp = workloads.svr_model_path(CACHE, genome, n_sv, seed=17, gamma=gamma, coef_scale=coef_scale)
    P = capi.make_params(140, 180, score_method=capi.SCORE_SVR)
    acc = capi.Accel(P)
    acc.load_model_file(mp)
    om = po.Model(mp)
    assert om.n_sv == n_sv
    regions = workloads.build_regions(acc, genome, ivs[:6], P, bwa_mode="hashed")
    grids, scores, records = acc.score_regions(regions, capi.SCORE_SVR)
    rng = np.random.default_rng(n_sv)
    worst = 0.0
    n_or = 60 if n_sv > 1000 else 300
    for ri, (rd, g) in enumerate(zip(regions, grids)):
        r = records[g.offset:g.offset + g.count]
        valid = np.nonzero((capi.rec_flags(r) & capi.FLAG_VALID) != 0)[0]
        pick = rng.choice(valid, size=400, replace=False)
        cands = [(ri,) + _decode(P, g, int(i))[1:] for i in pick]
        direct, _, _, _ = acc.score_candidates(cands, capi.SCORE_SVR)
        assert np.max(np.abs(direct - scores[g.offset + pick])) < 1e-8 * max(1.0, coef_scale)
        lrc = np.array(rd.c.long_range_content[:])
        for idx in pick[:n_or]:
            sk, d = po.design(P, rd, _decode(P, g, int(idx)))
            so, _, _ = po.score_designed(d, capi.SCORE_SVR, lrc, om)
            worst = max(worst, abs(scores[g.offset + idx] - so))
    assert worst < 1e-9 * max(1.0, coef_scale), worst                      # far inside the 1e-5 gate
    acc.close()

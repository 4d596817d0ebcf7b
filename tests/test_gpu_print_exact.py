"""GPU: printed digits of SVR scores.  The front end prints scores with 6 significant digits (mipgen.cpp:774); the dense kernel's score differs
from the reference's double by ~1e-13 (window-separable form, another summation order), so a score that sits on a midpoint between two
6-digit numbers could print another last digit.  libmipgen_accel re-scores exactly those candidates in the reference's own operation order
(svm.cpp:329-368, 2511-2515) - this test PLANTS such a score: rho is chosen so that the reference's double and the dense kernel's value fall on
opposite sides of a rounding midpoint, and the library must print the reference's digits."""
import os
import re

import numpy as np
import pytest

from mipgen_amd import capi, synth
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


def _midpoint_below_or_at(t):
    e10 = int(np.floor(np.log10(abs(t))))
    unit = 10.0 ** (e10 - 5)
    k = np.floor(abs(t) / unit)
    return (k + 0.5) * unit, unit


def test_planted_boundary_score_prints_the_reference_digits(tmp_path):
    genome = synth.random_genome(12000, 11)
    model_path = str(tmp_path / "m.model")
    synth.synthetic_svr_model(model_path, genome, 1024, seed=5, rho=-1.7)
    P = capi.make_params(130, 135, score_method=capi.SCORE_SVR, arm_pairs=synth.arm_pairs_from_sums([44, 45]))
    rd = capi.build_region(genome, "1", 5000, 5040, P, label="b", lrc=np.linspace(0.02, 0.25, 44))
    om = po.Model(model_path)
    acc = capi.Accel(P)
    acc.load_model_file(model_path)
    acc.set_print_exact(False)
    grids, s_dense, rec = acc.score_regions([rd], capi.SCORE_SVR)
    g = grids[0]
    _, s_or, _ = po.score_region_dense(P, rd, capi.SCORE_SVR, om)
    flags = capi.rec_flags(rec)
    ok = ((flags & capi.FLAG_VALID) != 0) & ((flags & capi.FLAG_GUARD) == 0) & (capi.rec_ext_copy(rec) > 0) & (capi.rec_lig_copy(rec) > 0) & (np.abs(s_or) > 0.1)
    diff = np.where(ok, np.abs(s_dense - s_or), 0.0)
    idx = int(np.argmax(diff))
    assert 1e-15 < diff[idx] < 1e-9, diff[idx]                     # the dense kernel is close, but not bit-identical: there is something to plant
    rho = om.rho if hasattr(om, "rho") else -1.7
    S_or, S_dense = s_or[idx] + rho, s_dense[idx] + rho
    B, unit = _midpoint_below_or_at(s_or[idx])
    rho2 = 0.5 * (S_or + S_dense) - B                              # the two values now straddle the midpoint B
    text = open(model_path).read()
    text2 = re.sub(r"^rho .*$", "rho " + repr(float(rho2)), text, count=1, flags=re.M)
    assert text2 != text
    model2 = str(tmp_path / "m2.model")
    open(model2, "w").write(text2)
    om2 = po.Model(model2)
    A = P.n_arm_pairs
    a = idx % A; row = idx // A; strand = row & 1; rest = row >> 1
    ki, pi = rest % g.n_sizes, rest // g.n_sizes
    cand = (0, g.first_pos + pi, P.max_capture_size - (g.first_size_index + ki) * P.capture_increment, P.arm_ext[a], P.arm_lig[a], int(strand))
    sk, d = po.design(P, rd, cand)
    assert not sk
    ref, _, _ = po.score_designed(d, capi.SCORE_SVR, np.array(rd.c.long_range_content[:]), om2)     # the reference's double
    want = format(ref, ".6g")
    # without the re-score the dense value prints another last digit ...
    acc.load_model_file(model2)
    _, s_off, _ = acc.score_regions([rd], capi.SCORE_SVR)
    assert (s_off[idx] - B) * (ref - B) < 0, (s_off[idx], ref, B)
    assert format(s_off[idx], ".6g") != want
    # ... with it (the default) the library prints the reference's digits, and the value is the reference's double to the last bits
    acc.set_print_exact(True)
    _, s_on, _ = acc.score_regions([rd], capi.SCORE_SVR)
    assert format(s_on[idx], ".6g") == want, (s_on[idx], ref, want)
    assert abs(s_on[idx] - ref) <= 2e-15 * max(1.0, abs(ref))
    # only candidates near a midpoint were touched, and every touched one moved by less than the kernels' error
    moved = np.nonzero(s_on != s_off)[0]
    assert idx in moved and moved.size < max(50, s_on.size // 500)
    assert np.all(np.abs(s_on[moved] - s_off[moved]) < 1e-9)
    # the list scorer of mixed designs (k_features_batch + k_svr_gemm) goes through the same fix
    cands = [cand] * 300
    sc, _, _, _ = acc.score_candidates(cands, capi.SCORE_SVR)
    assert format(sc[0], ".6g") == want and abs(sc[0] - ref) <= 2e-15 * max(1.0, abs(ref))
    acc.close()


def test_logistic_scores_near_a_print_midpoint_are_rescored_in_the_reference_order():
    """Round 5: the same guarantee for LOGISTIC scores.  k_logistic_dense regroups the 69 terms of the exponent by sequence window (error ~1e-14), so a
    score within that distance of a 6-digit rounding midpoint could print another last digit (mipgen.cpp:774); those candidates are re-scored with
    the terms in the reference's own order, every operation rounded on its own (SVMipv4.cpp:176-247), and overwritten.  A 5 kb region at 27 capture
    sizes holds 1.6e7 candidates: a few dozen sit that close to a midpoint; exactly those move, and where they move to is the reference's double."""
    from mipgen_amd import workloads
    genome = workloads.regions5k_genome()
    ivs = workloads.regions5k_intervals(1)
    P = capi.make_params(120, 250, score_method=capi.SCORE_LOGISTIC)
    regions = workloads.build_regions5k(None, genome, ivs, P, with_lrc=False)
    acc = capi.Accel(P)
    acc.set_print_exact(False)
    grids, s_off, rec = acc.score_regions(regions, capi.SCORE_LOGISTIC)
    acc.set_print_exact(True)
    _, s_on, _ = acc.score_regions(regions, capi.SCORE_LOGISTIC)
    g = grids[0]
    moved = np.nonzero((s_on != s_off) & ~(np.isnan(s_on) & np.isnan(s_off)))[0]
    assert 0 < moved.size < s_on.size // 10_000, moved.size
    assert np.all(np.abs(s_on[moved] - s_off[moved]) < 1e-11)
    A = P.n_arm_pairs
    rd = regions[0]
    n_exact = 0
    for idx in moved[:60]:
        a = int(idx % A); row = int(idx // A); strand = row & 1; rest = row >> 1
        ki, pi = rest % g.n_sizes, rest // g.n_sizes
        cand = (0, g.first_pos + pi, P.max_capture_size - (g.first_size_index + ki) * P.capture_increment, P.arm_ext[a], P.arm_lig[a], int(strand))
        sk, d = po.design(P, rd, cand)
        assert not sk
        ref, _, _ = po.score_designed(d, capi.SCORE_LOGISTIC, np.zeros(44), None)            # the reference's double (oracle: bit-exact, tests/test_oracle_golden.py)
        # the re-scored value is the reference's to the last bits (pow of the device math library vs glibc's: < 4 ulp), the dense value was ~1e-14 off
        assert abs(s_on[idx] - ref) <= 4 * np.spacing(abs(ref)), (s_on[idx], ref)
        assert format(s_on[idx], ".6g") == format(ref, ".6g")
        n_exact += s_on[idx] == ref
    assert n_exact > 0
    acc.close()


@pytest.mark.parametrize("route", ["dense", "fallback"])
def test_saturated_logistic_scores_are_the_references_doubles(route):
    """Inside a (CCG)n run the logistic exponent reaches 36.7-37.4: b^x lies in [2^53, 2^54), 1 + y is a tie the reference rounds to even on the LAST
    bit of its pow, and its score - exactly 1.0, or one / two ulps below - decides the strict comparisons of collapse / condense and the (int)
    truncations of mipgen.cpp:494-497 (found by the differential probe on the hard genome; goldens design_hard_saturated_*).  The dense kernel takes the
    correctly rounded quotient from 2^20 on (1 + y is exact below 2^53: its 1e-13 on y moves nothing) and lists the candidates of the tie binade, which are
    re-scored in the reference's term order with a correctly rounded power before anything is replayed: every score of the band is the reference's double."""
    from tests import helpers as H
    meta = H.load_design("hard_saturated_logistic")
    P = H.design_params(meta)
    if route == "fallback":
        # capture sizes 50 apart: the window tables of k_logistic_dense do not fit its LDS budget, k_records_logistic<true> scores the batch (and lists
        # the tie binade just the same)
        P = capi.make_params(120, 270, capture_increment=50, arm_pairs=synth.arm_pairs_from_sums(meta["sums"]))
    genome = H.golden_genome("genome4_chr4.fa.gz").upper()
    regions = H.design_regions(dict(meta, minC=P.min_capture_size), genome, P)
    acc = capi.Accel(P)
    grids, scores, records = acc.score_regions(regions, capi.SCORE_LOGISTIC)
    g = grids[0]
    _, o, _ = po.score_region_dense(P, regions[0], capi.SCORE_LOGISTIC, None)
    s = np.asarray(scores[:g.count]); o = np.asarray(o)
    band = np.isfinite(o) & (o >= 1.0 - 2.0 ** -20)
    assert int(band.sum()) > (4000 if route == "dense" else 500) and int((o == 1.0).sum()) > (3000 if route == "dense" else 100)   # the region really saturates
    tie = band & (o >= 1.0 - 3 * 2.0 ** -53) & (o < 1.0)                                  # one / two ulps below 1.0: only the tie binade produces these
    assert int(tie.sum()) >= (10 if route == "dense" else 1)
    bad = np.flatnonzero(band & (s.view(np.int64) != o.view(np.int64)))
    assert bad.size == 0, (int(bad.size), [(int(k), float(o[k]), float(s[k])) for k in bad[:5]])
    assert np.array_equal(s == 1.0, o == 1.0)
    # the list route (mipgen_accel_score_candidates: k_candidates) on the tie binade alone
    A = P.n_arm_pairs
    cl = []
    for k in np.flatnonzero(tie):
        a = int(k % A); row = int(k // A); rest = row >> 1
        cl.append((0, g.first_pos + rest // g.n_sizes, P.max_capture_size - (g.first_size_index + rest % g.n_sizes) * P.capture_increment, P.arm_ext[a], P.arm_lig[a], row & 1))
    sl, _, _, _ = acc.score_candidates(cl, capi.SCORE_LOGISTIC)
    assert np.array_equal(np.asarray(sl).view(np.int64), o[tie].view(np.int64))
    acc.close()

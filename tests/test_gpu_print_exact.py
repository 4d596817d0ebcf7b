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

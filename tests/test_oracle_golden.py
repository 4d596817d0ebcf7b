"""CPU: the oracle (plain-C restatement) against the committed golden vectors, which were produced by the
real reference (tests/golden/make_golden.py).  Bit-exact everywhere: same arithmetic, same order."""
import json
import os

import numpy as np
import pytest

from mipgen_amd import capi
from oracle import pyoracle as po
from tests import helpers as H


@pytest.fixture(scope="module", params=["candidates", "candidates_hard"])
def cands(request):
    """candidates: iid ACGT + N from the first golden genome; candidates_hard: cut from the hard genome (ambiguity codes, '-' bytes, lower case left
    as it is in every fifth, homopolymers, microsatellites, 20 % / 70 % GC) with ambiguity codes and '-' planted in arms and inserts."""
    with open(os.path.join(H.GOLDEN, request.param + ".json")) as fh:
        meta = json.load(fh)
    z = np.load(os.path.join(H.GOLDEN, request.param + ".npz"))
    meta["name"] = request.param
    return meta, z


def test_logistic_and_parameters_bit_exact(cands):
    meta, z = cands
    n_guard = n_nan = 0
    for i, c in enumerate(meta["candidates"]):
        e, l, s = po.orient(c["strand"], c["ext_fwd"].encode(), c["lig_fwd"].encode(), c["ins_fwd"].encode())
        sc, ints = po.get_score(e, l, s, c["ext_copy"], c["lig_copy"])
        ref = z["logistic"][i]
        assert sc == ref or (np.isnan(sc) and np.isnan(ref)), (i, sc, ref)
        x = po.get_parameters(e, l, s, c["ext_copy"], c["lig_copy"], z["lrc"][i])
        assert np.array_equal(x, z["params"][i], equal_nan=True), i
        n_guard += sc == -1000.0
        n_nan += bool(np.isnan(sc))
        # integer fields are consistent with the reference's frequency features
        if sc != -1000.0:
            assert x[0] * len(e) == pytest.approx(ints.ext_a, abs=1e-9)
            assert x[151] == ints.scan_size
    assert n_guard > 20 and (n_nan > 5 or meta["name"] == "candidates_hard")      # the edge cases are really in the fixture
    if meta["name"] == "candidates_hard":
        odd = sum(1 for c in meta["candidates"] if any(ch not in "ACGTN" for ch in c["ext_fwd"] + c["lig_fwd"]))
        assert odd > 60, odd                                                       # arms with bytes other than A C G T N


def test_svr_predict_bit_exact(cands):
    meta, z = cands
    for name, key in (("svr_syn_64.model", "svr64"), ("svr_syn_200.model", "svr200")):
        m = po.Model(os.path.join(H.GOLDEN, "models", name))
        for i in range(len(meta["candidates"])):
            got = m.predict(z["params"][i])
            assert got == z[key][i] or (np.isnan(got) and np.isnan(z[key][i])), (name, i)


def test_model_written_by_libsvm_itself():
    """tests/golden/models/svr_libsvm_trained.model was TRAINED and WRITTEN by the reference's own libsvm (svm_train, svm.cpp:2095; svm_save_model,
    svm.cpp:2644-2757; oracle/ref_driver.cpp: ref_svm_train_save) - genuine svm_save_model output, not this repository's writer: the oracle's loader
    reads the header svm_load_model read (svm.cpp:2779-2899) and its predictions equal the reference's svm_predict on 150 candidates, edge cases included."""
    z = np.load(os.path.join(H.GOLDEN, "libsvm_trained.npz"))
    m = po.Model(os.path.join(H.GOLDEN, "models", "svr_libsvm_trained.model"))
    assert m.n_sv == int(z["n_sv"][0]) and m.gamma == float(z["gamma"][0]) and m.rho == float(z["rho"][0])
    n_inf = 0
    for i in range(z["params"].shape[0]):
        got = m.predict(z["params"][i])
        assert got == z["svr"][i] or (np.isnan(got) and np.isnan(z["svr"][i])), i
        n_inf += bool(np.isinf(z["params"][i]).any())
    assert n_inf >= 5 and (z["params"] == 0).all(axis=1).sum() >= 5       # log10(0) and all-zero guard vectors are among them


def test_long_range_content_bit_exact(cands):
    meta, z = cands
    g = H.golden_genome() if meta["name"] == "candidates" else H.golden_genome("genome4_chr4.fa.gz")
    for i, lr in enumerate(meta["long_range"]):
        seq = (g if meta["name"] == "candidates" or lr["raw"] else g.upper())[lr["offset"]:lr["offset"] + lr["len"]]
        got = po.long_range_content(seq, lr["chrom_seq_start"], lr["chrom_seq_stop"])
        assert np.array_equal(got, z["lr_out"][i])


def test_model_loader_rejects_missing():
    with pytest.raises(RuntimeError):
        po.Model("/nonexistent/mipgen_svr.model")


@pytest.mark.parametrize("name", ["logistic_snp_trf", "svr_small", "mixed_small"])
def test_enumeration_matches_reference_all_mips(name):
    """Literal loop restatement + print_details == the reference's all_mips.txt, byte for byte
    (emitted set, order, mip_name numbering, copies, flags, SNP counts, 6-digit scores)."""
    meta = H.load_design(name)
    g = H.golden_genome()
    P = H.design_params(meta)
    model = po.Model(os.path.join(H.GOLDEN, "models", meta["model"])) if meta["model"] else None
    regions = H.design_regions(meta, g, P, lrc_fn=po.long_range_content)
    ref = H.ref_lines(meta, "all_mips")[1:]
    middle = H.middle_of(meta["tags"])
    method = 1 if meta["method"] == "svr" else 0
    k = 0
    for rd in regions:
        n, buf = po.enumerate_region(P, rd, method, model)
        for i in range(n):
            em = buf[i]
            _, d = po.design(P, rd, (0, em.scan_start, em.capture_size, em.ext_len, em.lig_len, em.strand))
            line = po.print_details(rd, em.strand, d, em.score, middle, k + 1)
            assert line.rstrip(b"\n") == H.normalise_flags(ref[k]), (name, k)
            k += 1
    assert k == len(ref)


@pytest.mark.parametrize("name", ["logistic_snp_trf", "svr_small", "mixed_small", "logistic_default_arms"])
def test_dense_grid_plus_replay_equals_literal_loop(name):
    """Scoring the dense grid and replaying the early exits gives exactly the literal loop's emitted list."""
    meta = H.load_design(name)
    g = H.golden_genome()
    P = H.design_params(meta)
    model = po.Model(os.path.join(H.GOLDEN, "models", meta["model"])) if meta["model"] else None
    method = 1 if meta["method"] == "svr" else 0
    total = 0
    for rd in H.design_regions(meta, g, P, lrc_fn=po.long_range_content):
        grid, scores, records = po.score_region_dense(P, rd, method, model)
        n_emit, mask = po.replay_region(P, rd, scores, records)
        n, buf = po.enumerate_region(P, rd, method, model)
        assert n == n_emit
        idx = np.nonzero(mask)[0]
        lit = np.array([buf[i].dense_index for i in range(n)], dtype=np.int64)      # generation order (plus, minus per pair)
        assert np.array_equal(idx, np.sort(lit)) and np.unique(lit).size == n        # same set; the dense order is strand-major
        lit_scores = np.array([buf[i].score for i in range(n)])
        assert np.array_equal(scores[lit], lit_scores, equal_nan=True)
        total += n
    assert total == meta["lines"]["all_mips"] - 1

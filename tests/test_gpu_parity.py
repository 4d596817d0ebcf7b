"""GPU parity tests: the HIP path, called through the C-ABI (libmipgen_accel.so), against the oracle and the
committed golden vectors.  Integer fields bit-exact; logistic / SVR scores within 1e-5 (BASELINE.json north_star).

Run on the MI355X box:  python -m pytest tests -m gpu -x -q
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

from mipgen_amd import capi, synth
from oracle import pyoracle as po
from tests import helpers as H

pytestmark = pytest.mark.gpu
TOL = 1e-5          # score tolerance stated by BASELINE.json north_star


def _close(a, b, tol=TOL):
    a, b = np.asarray(a), np.asarray(b)
    nan_ok = np.isnan(a) & np.isnan(b)
    inf_ok = np.isinf(a) & (a == b)
    with np.errstate(invalid="ignore"):
        d = np.abs(a - b)
    ok = nan_ok | inf_ok | (d <= tol)
    return ok, np.nanmax(np.where(nan_ok | inf_ok, 0.0, d)) if a.size else 0.0


@pytest.fixture(scope="module")
def genome():
    return H.golden_genome()


def _genome_of(meta, genome):
    """The chromosome a golden design is laid on, as the reference's input stage hands it on (upper case, mipgen.cpp:1208); the hard genome's file
    holds lower case, ambiguity codes and '-' bytes."""
    name = meta.get("genome", "genome_chr1.fa.gz")
    return genome if name == "genome_chr1.fa.gz" else H.golden_genome(name).upper()


HARD = ["hard_logistic", "hard_svr", "hard_mixed"]     # the hard genome: ambiguity codes, lower case, '-', homopolymers, microsatellites, GC 20 / 70 %


def _model_path(meta):
    return os.path.join(H.GOLDEN, "models", meta["model"]) if meta["model"] else os.path.join(H.GOLDEN, "models", "svr_syn_64.model")


@pytest.mark.parametrize("name", ["logistic_snp_trf", "svr_small", "mixed_small", "logistic_default_arms"] + HARD)
@pytest.mark.parametrize("method", [capi.SCORE_LOGISTIC, capi.SCORE_SVR])
def test_dense_grid_vs_oracle(name, method, genome):
    """Every dense-grid candidate of the golden designs: records bit-exact, scores within 1e-5."""
    meta = H.load_design(name)
    P = H.design_params(meta)
    regions = H.design_regions(meta, _genome_of(meta, genome), P, lrc_fn=po.long_range_content)
    mp = _model_path(meta)
    acc = capi.Accel(P)
    acc.load_model_file(mp)
    om = po.Model(mp)
    grids, scores, records = acc.score_regions(regions, method)
    n_guard = 0
    for rd, g in zip(regions, grids):
        og, os_, or_ = po.score_region_dense(P, rd, method, om)
        assert (g.first_pos, g.n_pos, g.first_size_index, g.n_sizes, g.count) == (og.first_pos, og.n_pos, og.first_size_index, og.n_sizes, og.count)
        s = scores[g.offset:g.offset + g.count]
        r = records[g.offset:g.offset + g.count]
        bad = np.nonzero(r != or_)[0]
        assert bad.size == 0, (name, "first differing record", int(bad[0]), hex(int(r[bad[0]])), hex(int(or_[bad[0]])))
        ok, mx = _close(s, os_)
        assert ok.all(), (name, method, "max |diff|", mx, "at", int(np.nonzero(~ok)[0][0]))
        if method == capi.SCORE_LOGISTIC:
            guard = (capi.rec_flags(r) & capi.FLAG_GUARD) != 0
            assert np.all(s[guard & ((capi.rec_flags(r) & capi.FLAG_VALID) != 0)] == -1000.0)
            n_guard += int(guard.sum())
    acc.close()


def _region_for_candidate(c, lrc, P):
    """Embed one golden known-answer candidate (raw forward-strand strings) in a synthetic region."""
    ext, lig, ins = c["ext_fwd"].encode(), c["lig_fwd"].encode(), c["ins_fwd"].encode()
    pad = b"ACGT" * 20
    up, down = (ext, lig) if c["strand"] == 0 else (lig, ext)
    seq = pad + up + ins + down + pad
    seq_start = 1000
    p = seq_start + len(pad) + len(up)
    n = len(seq)
    copy = {}
    for ln in {len(ext), len(lig)}:
        copy[ln] = np.ones(n, dtype=np.int32)
    ext_start = p - len(ext) if c["strand"] == 0 else p + len(ins)
    lig_start = p + len(ins) if c["strand"] == 0 else p - len(lig)
    # when both arms have the same length and the same start cannot happen; distinct starts always
    copy[len(ext)][ext_start - seq_start] = c["ext_copy"]
    copy[len(lig)][lig_start - seq_start] = c["lig_copy"]
    rd = capi.RegionData(p, p + 10, seq_start, seq, copy=copy, lrc=lrc)
    cand = (0, p, len(ext) + len(lig) + len(ins), len(ext), len(lig), c["strand"])
    return rd, cand


def test_sparse_candidates_vs_golden_known_answers():
    """mipgen_accel_score_candidates against the reference's own known answers (tests/golden/candidates.*):
    the 192 features bit-exact (integer counts / integer denominators), logistic and SVR within 1e-5."""
    _known_answers("candidates", 100)


def test_sparse_candidates_vs_known_answers_on_hard_sequence():
    """The same against tests/golden/candidates_hard.*: candidates cut from the hard genome - ambiguity codes (R Y M K S W B D H V) and '-' bytes in
    arms AND inserts (at the ligation junction, at the arm ends), bytes left in lower case, homopolymers, microsatellites, 20 % / 70 % GC.  The
    reference guards on N and '-' only (SVMipv4.cpp:116); any other byte simply is not A / C / G / T to its counters (:118-141) and passes through
    reverse_comp (MinusSVMipv4.cpp:24-25)."""
    _known_answers("candidates_hard", 200)


def _known_answers(fixture, n_min):
    with open(os.path.join(H.GOLDEN, fixture + ".json")) as fh:
        meta = json.load(fh)
    z = np.load(os.path.join(H.GOLDEN, fixture + ".npz"))
    n_checked = n_odd = 0
    for model_name, key in (("svr_syn_64.model", "svr64"), ("svr_syn_200.model", "svr200")):
        for i, c in enumerate(meta["candidates"]):
            n_odd += any(ch not in "ACGTN" for ch in c["ext_fwd"] + c["lig_fwd"])
            if i % 2 == (0 if key == "svr64" else 1) and i > 60:
                continue                                    # alternate candidates between the two models to bound run time
            e, l = len(c["ext_fwd"]), len(c["lig_fwd"])
            P = capi.make_params(100, 400, score_method=capi.SCORE_SVR, arm_pairs=[(e, l)])
            rd, cand = _region_for_candidate(c, z["lrc"][i], P)
            acc = capi.Accel(P)
            acc.load_model_file(os.path.join(H.GOLDEN, "models", model_name))
            acc.upload([rd])
            s_svr, rec, feats, ints = acc.score_candidates([cand], capi.SCORE_SVR, want_features=True, want_ints=True)
            s_log, _, _, _ = acc.score_candidates([cand], capi.SCORE_LOGISTIC)
            acc.close()
            assert np.array_equal(feats[0], z["params"][i], equal_nan=True), (i, np.nonzero(feats[0] != z["params"][i]))
            ok, mx = _close(s_log, [z["logistic"][i]])
            assert ok.all(), ("logistic", i, s_log[0], z["logistic"][i])
            ok, mx = _close(s_svr, [z[key][i]])
            assert ok.all(), ("svr", key, i, s_svr[0], z[key][i])
            # integer fields against the oracle's restatement
            oe, ol, oi = po.orient(c["strand"], c["ext_fwd"].encode(), c["lig_fwd"].encode(), c["ins_fwd"].encode())
            _, oints = po.get_score(oe, ol, oi, c["ext_copy"], c["lig_copy"])
            for f in ("ext_a", "ext_c", "ext_g", "ext_t", "lig_a", "lig_c", "lig_g", "lig_t", "ins_a", "ins_c", "ins_g", "ins_t",
                      "run_count", "junction", "ext_copy", "lig_copy", "scan_size"):
                assert getattr(ints[0], f) == getattr(oints, f), (i, f)
            n_checked += 1
    assert n_checked > n_min and (fixture == "candidates" or n_odd > 60)


@pytest.mark.parametrize("fixture", ["candidates", "candidates_hard"])
def test_long_range_content_vs_golden(fixture, genome):
    with open(os.path.join(H.GOLDEN, fixture + ".json")) as fh:
        meta = json.load(fh)
    z = np.load(os.path.join(H.GOLDEN, fixture + ".npz"))
    g = genome if fixture == "candidates" else H.golden_genome("genome4_chr4.fa.gz")
    P = capi.make_params(120, 130)
    acc = capi.Accel(P)
    for i, lr in enumerate(meta["long_range"]):
        seq = (g if fixture == "candidates" or lr["raw"] else g.upper())[lr["offset"]:lr["offset"] + lr["len"]]
        got = acc.long_range_content(seq, lr["chrom_seq_start"], lr["chrom_seq_stop"])
        assert np.array_equal(got, z["lr_out"][i]), (fixture, i)       # integer counts / integer denominator: bit-exact
    acc.close()


@pytest.mark.parametrize("name", ["logistic_snp_trf", "svr_small", "mixed_small", "logistic_default_arms"] + HARD)
def test_replay_condense_vs_oracle(name, genome):
    """Device replay of the early exits + condense fold == the oracle's, fed with the device's own scores."""
    meta = H.load_design(name)
    P = H.design_params(meta)
    regions = H.design_regions(meta, _genome_of(meta, genome), P, lrc_fn=po.long_range_content)
    method = capi.SCORE_SVR if meta["method"] == "svr" else capi.SCORE_LOGISTIC
    acc = capi.Accel(P)
    if meta["model"]:
        acc.load_model_file(_model_path(meta))
    grids, scores, records = acc.score_regions(regions, method)
    acc.replay_condense()
    emitted, surv, mask = acc.download_replay()
    pos0 = 0
    total = 0
    for ri, (rd, g) in enumerate(zip(regions, grids)):
        s = scores[g.offset:g.offset + g.count]
        r = records[g.offset:g.offset + g.count]
        n_emit, omask = po.replay_region(P, rd, s, r)
        assert emitted[ri] == n_emit
        assert np.array_equal(mask[g.offset:g.offset + g.count], omask)
        osurv = po.condense_region(P, rd, s, r, omask)
        got = surv[2 * pos0:2 * (pos0 + g.n_pos)]
        exp_idx = np.where(osurv["cand_index"] >= 0, osurv["cand_index"] + g.offset, -1)
        assert np.array_equal(got["cand_index"], exp_idx)
        assert np.array_equal(got["score"], osurv["score"], equal_nan=True)
        assert np.array_equal(got["record"], osurv["record"])
        pos0 += g.n_pos
        total += n_emit
    # with the device's scores the emitted count equals the reference's all_mips line count
    assert total == meta["lines"]["all_mips"] - 1
    acc.close()


@pytest.mark.parametrize("name", ["logistic_snp_trf", "svr_small", "mixed_small"] + HARD)
def test_all_oracle_chain_vs_device_chain(name, genome):
    """The chain closed on BOTH sides: the oracle's own dense scores -> its replay -> its condense fold, against the device's scores -> replay ->
    fold.  (The test above feeds the oracle's control flow the device's scores; here nothing crosses.)  Candidate indices, records and emitted
    masks must be identical - the two score sets differ by ~1e-14, which flips a comparison only on an exact tie -, scores within 1e-5."""
    meta = H.load_design(name)
    P = H.design_params(meta)
    regions = H.design_regions(meta, _genome_of(meta, genome), P, lrc_fn=po.long_range_content)
    method = capi.SCORE_SVR if meta["method"] == "svr" else capi.SCORE_LOGISTIC
    om = po.Model(_model_path(meta)) if meta["model"] and method == capi.SCORE_SVR else None
    acc = capi.Accel(P)
    if meta["model"]:
        acc.load_model_file(_model_path(meta))
    grids, scores, records = acc.score_regions(regions, method)
    acc.replay_condense()
    emitted, surv, mask = acc.download_replay()
    pos0 = 0
    for ri, (rd, g) in enumerate(zip(regions, grids)):
        _, os_, or_ = po.score_region_dense(P, rd, method, om)
        n_emit, omask = po.replay_region(P, rd, os_, or_)
        osurv = po.condense_region(P, rd, os_, or_, omask)
        assert emitted[ri] == n_emit
        assert np.array_equal(mask[g.offset:g.offset + g.count], omask)
        got = surv[2 * pos0:2 * (pos0 + g.n_pos)]
        assert np.array_equal(got["cand_index"], np.where(osurv["cand_index"] >= 0, osurv["cand_index"] + g.offset, -1))
        assert np.array_equal(got["record"], osurv["record"])
        both_nan = np.isnan(got["score"]) & np.isnan(osurv["score"])
        assert np.all(both_nan | (np.abs(got["score"] - osurv["score"]) <= 1e-5))
        pos0 += g.n_pos
    acc.close()


def test_edge_cases(genome):
    """Empty batch, ragged capture-size sets (static skip), the widest capture sweep (120-250, K=27), regions at the
    chromosome start (bounds skips), copy 0 / 100 / 101, N runs."""
    P = capi.make_params(120, 250, score_method=capi.SCORE_SVR)
    mp = os.path.join(H.GOLDEN, "models", "svr_syn_64.model")
    acc = capi.Accel(P)
    acc.load_model_file(mp)
    om = po.Model(mp)
    # empty batch
    grids, scores, records = acc.score_regions([], capi.SCORE_LOGISTIC)
    assert len(grids) == 0 and scores.size == 0
    # ragged: a 40-bp region keeps few sizes, a 260-bp one keeps all 27; one region hugging the chromosome start
    regs = [capi.build_region(genome, "1", 3000, 3040, P, bwa_mode="hashed", label="short", lrc=np.full(44, 0.1)),
            capi.build_region(genome, "1", 6000, 6260, P, bwa_mode="hashed", label="long", lrc=np.full(44, 0.2)),
            capi.build_region(genome, "1", 150, 200, P, bwa_mode="hashed", label="edge", lrc=np.full(44, 0.05))]
    for method in (capi.SCORE_LOGISTIC, capi.SCORE_SVR):
        grids, scores, records = acc.score_regions(regs, method)
        assert grids[0].n_sizes < grids[1].n_sizes == 27
        rng = np.random.default_rng(3)
        for rd, g in zip(regs, grids):
            _, os_, or_ = po.score_region_dense(P, rd, capi.SCORE_LOGISTIC, om)        # records (method independent)
            r = records[g.offset:g.offset + g.count]
            assert np.array_equal(r, or_)
            if method == capi.SCORE_LOGISTIC:
                ok, mx = _close(scores[g.offset:g.offset + g.count], os_)
                assert ok.all(), mx
            else:
                # the oracle's SVR is slow: check a random sample of 1500 candidates + every guard / zero-copy one
                valid = np.nonzero((capi.rec_flags(r) & capi.FLAG_VALID) != 0)[0]
                special = valid[((capi.rec_flags(r[valid]) & capi.FLAG_GUARD) != 0) | (capi.rec_ext_copy(r[valid]) == 0) |
                                (capi.rec_lig_copy(r[valid]) == 0)][:300]
                pick = np.unique(np.concatenate([rng.choice(valid, size=min(1500, valid.size), replace=False), special]))
                A = P.n_arm_pairs
                for idx in pick:
                    a = idx % A
                    row = idx // A
                    strand = row & 1
                    rest = row >> 1
                    ki, pi = rest % g.n_sizes, rest // g.n_sizes
                    cand = (0, g.first_pos + pi, P.max_capture_size - (g.first_size_index + ki) * P.capture_increment,
                            P.arm_ext[a], P.arm_lig[a], int(strand))
                    sk, d = po.design(P, rd, cand)
                    assert not sk
                    so, _, _ = po.score_designed(d, capi.SCORE_SVR, np.array(rd.c.long_range_content[:]), om)
                    ok, mx = _close([scores[g.offset + idx]], [so])
                    assert ok.all(), (rd.label, cand, scores[g.offset + idx], so)
        # invalid (bounds-skipped) candidates exist only in the chromosome-start region
        assert ((capi.rec_flags(records[grids[2].offset:grids[2].offset + grids[2].count]) & capi.FLAG_VALID) == 0).any()
    acc.close()


def test_error_paths(genome):
    lib = capi.load_library()
    P = capi.make_params(120, 130)
    acc = capi.Accel(P)
    with pytest.raises(capi.AccelError):
        acc.load_model_file("/nonexistent/mipgen_svr.model")          # the reference would segfault (svm.cpp:2507)
    rd = capi.build_region(genome, "1", 5000, 5050, P)
    acc.upload([rd])
    with pytest.raises(capi.AccelError):
        acc.score_resident(capi.SCORE_SVR)                              # no model loaded
    with pytest.raises(capi.AccelError):
        acc.download()                                                  # nothing scored
    bad = capi.make_params(120, 130)
    bad.abi_version = 99
    with pytest.raises(capi.AccelError):
        capi.Accel(bad)
    acc.close()
    assert lib.mipgen_accel_device_count() >= 1


def test_very_wide_capture_range(genome):
    """capture 120-300 step 20 (scan sizes up to 260, nine sizes spanning 160 bases): the host shortens the capture-size runs
    until the tile fits LDS, the scan units take their long-array path; sampled candidates still match the oracle."""
    P = capi.make_params(120, 300, score_method=capi.SCORE_SVR, capture_increment=20)
    mp = os.path.join(H.GOLDEN, "models", "svr_syn_64.model")
    acc = capi.Accel(P)
    acc.load_model_file(mp)
    om = po.Model(mp)
    rd = capi.build_region(genome, "1", 8000, 8320, P, bwa_mode="hashed", label="wide", lrc=np.full(44, 0.07))
    grids, scores, records = acc.score_regions([rd], capi.SCORE_SVR)
    g = grids[0]
    valid = np.nonzero((capi.rec_flags(records) & capi.FLAG_VALID) != 0)[0]
    rng = np.random.default_rng(9)
    A = P.n_arm_pairs
    for idx in rng.choice(valid, size=400, replace=False):
        a = idx % A; row = idx // A; strand = row & 1; rest = row >> 1
        ki, pi = rest % g.n_sizes, rest // g.n_sizes
        cand = (0, g.first_pos + pi, P.max_capture_size - (g.first_size_index + ki) * P.capture_increment, P.arm_ext[a], P.arm_lig[a], int(strand))
        sk, d = po.design(P, rd, cand)
        so, _, _ = po.score_designed(d, capi.SCORE_SVR, np.array(rd.c.long_range_content[:]), om)
        ok, _ = _close([scores[idx]], [so])
        assert ok.all(), (cand, scores[idx], so)
    acc.close()


def test_fragmented_arm_pair_order(genome):
    """Arm pairs in an order where consecutive pairs never share an arm-length sum (every candidate step opens a new insert
    factor list) and whose count does not fill the last chunk: the dense kernel must not depend on the reference's sum-sorted
    enumeration order."""
    base = capi.arm_pairs_of(capi.make_params(140, 160))
    rng = np.random.default_rng(4)
    pairs = [base[i] for i in rng.permutation(len(base))][:23]
    for i in range(1, len(pairs)):                      # break up accidental runs of equal sums
        if sum(pairs[i]) == sum(pairs[i - 1]):
            for j in range(i + 1, len(pairs)):
                if sum(pairs[j]) != sum(pairs[i - 1]) and (i + 1 >= len(pairs) or sum(pairs[j]) != sum(pairs[i + 1])):
                    pairs[i], pairs[j] = pairs[j], pairs[i]
                    break
    P = capi.make_params(140, 160, score_method=capi.SCORE_SVR, arm_pairs=pairs)
    mp = os.path.join(H.GOLDEN, "models", "svr_syn_64.model")
    acc = capi.Accel(P)
    acc.load_model_file(mp)
    om = po.Model(mp)
    rd = capi.build_region(genome, "1", 12000, 12090, P, bwa_mode="hashed", label="frag", lrc=np.full(44, 0.05))
    grids, scores, records = acc.score_regions([rd], capi.SCORE_SVR)
    g = grids[0]
    valid = np.nonzero((capi.rec_flags(records) & capi.FLAG_VALID) != 0)[0]
    A = P.n_arm_pairs
    for idx in rng.choice(valid, size=min(500, len(valid)), replace=False):
        a = idx % A; row = idx // A; strand = row & 1; rest = row >> 1
        ki, pi = rest % g.n_sizes, rest // g.n_sizes
        cand = (0, g.first_pos + pi, P.max_capture_size - (g.first_size_index + ki) * P.capture_increment, P.arm_ext[a], P.arm_lig[a], int(strand))
        sk, d = po.design(P, rd, cand)
        so, _, _ = po.score_designed(d, capi.SCORE_SVR, np.array(rd.c.long_range_content[:]), om)
        ok, _ = _close([scores[idx]], [so])
        assert ok.all(), (cand, scores[idx], so)
    acc.close()


@pytest.mark.parametrize("n_pairs", [3, 9, 70, 130])
def test_arm_pair_counts(genome, n_pairs):
    """Thread geometry of the dense SVR kernel: 3 and 9 pairs take the 5- and 10-step instantiations, 70 and 130 pairs need
    more chunks than 16 wavefronts hold at four waves per chunk (3 resp. 1 waves per chunk, fewer positions per tile)."""
    grid = sorted(((e, l) for e in range(16, 31) for l in range(18, 31)), key=lambda t: (-(t[0] + t[1]), t[0]))
    pairs = grid[20:20 + n_pairs]
    P = capi.make_params(150, 165, score_method=capi.SCORE_SVR, arm_pairs=pairs)
    mp = os.path.join(H.GOLDEN, "models", "svr_syn_64.model")
    acc = capi.Accel(P)
    acc.load_model_file(mp)
    om = po.Model(mp)
    rd = capi.build_region(genome, "1", 15000, 15060, P, bwa_mode="hashed", label="geom", lrc=np.full(44, 0.03))
    grids, scores, records = acc.score_regions([rd], capi.SCORE_SVR)
    g = grids[0]
    valid = np.nonzero((capi.rec_flags(records) & capi.FLAG_VALID) != 0)[0]
    rng = np.random.default_rng(n_pairs)
    A = P.n_arm_pairs
    for idx in rng.choice(valid, size=min(300, len(valid)), replace=False):
        a = idx % A; row = idx // A; strand = row & 1; rest = row >> 1
        ki, pi = rest % g.n_sizes, rest // g.n_sizes
        cand = (0, g.first_pos + pi, P.max_capture_size - (g.first_size_index + ki) * P.capture_increment, P.arm_ext[a], P.arm_lig[a], int(strand))
        sk, d = po.design(P, rd, cand)
        so, _, _ = po.score_designed(d, capi.SCORE_SVR, np.array(rd.c.long_range_content[:]), om)
        ok, _ = _close([scores[idx]], [so])
        assert ok.all(), (n_pairs, cand, scores[idx], so)
    acc.close()


def test_sv_split_variants(genome):
    """A dense SVR launch may be cut along the support-vector list (work unit = tile x SV part; partial sums added in part order by
    k_svr_finish).  Every split agrees with the oracle within 1e-5 and with the unsplit launch to rounding; a given split is bitwise
    reproducible."""
    meta = H.load_design("svr_small")
    P = H.design_params(meta)
    regions = H.design_regions(meta, genome, P, lrc_fn=po.long_range_content)
    mp = os.path.join(H.GOLDEN, "models", "svr_syn_200.model")
    acc = capi.Accel(P)
    acc.load_model_file(mp)
    om = po.Model(mp)
    ref = None
    for split in (1, 2, 3, 5):
        acc.set_sv_split(split)
        grids, scores, records = acc.score_regions(regions, capi.SCORE_SVR)
        _, again, _ = acc.score_regions(regions, capi.SCORE_SVR)
        assert np.array_equal(scores, again, equal_nan=True)
        if ref is None:
            ref = scores
            for rd, g in zip(regions, grids):
                _, os_, or_ = po.score_region_dense(P, rd, capi.SCORE_SVR, om)
                ok, mx = _close(scores[g.offset:g.offset + g.count], os_)
                assert ok.all(), mx
                assert np.array_equal(records[g.offset:g.offset + g.count], or_)
        else:
            assert np.nanmax(np.abs(scores - ref)) < 1e-11, split
    acc.close()


def test_result_windows(genome):
    """A batch whose dense results exceed the result arrays is scored window by window (consecutive regions); inputs stay resident.
    Dense results, replay masks and survivors equal the single-window run; mipgen_accel_score_condense_all leaves the survivors of
    every window behind."""
    meta = H.load_design("logistic_default_arms")
    P = H.design_params(meta)
    regions = H.design_regions(meta, genome, P)
    acc = capi.Accel(P)
    grids, s1, r1 = acc.score_regions(regions, capi.SCORE_LOGISTIC)
    assert acc.window_count() == 1
    acc.replay_condense()
    e1, v1, m1 = acc.download_replay()
    g1, s1c, r1c = acc.score_regions_one_call(regions, capi.SCORE_LOGISTIC, capacity=int(s1.size) + 5)
    assert np.array_equal(s1, s1c, equal_nan=True) and np.array_equal(r1, r1c)
    acc.set_window_candidates(int(max(g.count for g in grids)) + 1)              # one region per window
    grids2, s2, r2 = acc.score_regions(regions, capi.SCORE_LOGISTIC)
    assert acc.window_count() == len(regions) > 1
    assert [g.offset for g in grids2] == [g.offset for g in grids]
    assert np.array_equal(s1, s2, equal_nan=True) and np.array_equal(r1, r2)
    with pytest.raises(capi.AccelError):
        acc.score_resident(capi.SCORE_LOGISTIC)                                   # several windows: the caller must say which
    pos0 = 0
    for w in range(acc.window_count()):
        wi = acc.window_info(w)
        acc.score_window(w, capi.SCORE_LOGISTIC)
        acc.replay_condense()
        e, v, m = acc.download_replay(window=w)
        assert np.array_equal(e, e1[wi["first_region"]:wi["first_region"] + wi["n_regions"]])
        assert np.array_equal(m, m1[wi["first_candidate"]:wi["first_candidate"] + wi["n_candidates"]])
        ref = v1[2 * pos0:2 * (pos0 + wi["n_positions"])]
        for f in ("cand_index", "record"):
            assert np.array_equal(v[f], ref[f]), (w, f)
        assert np.array_equal(v["score"], ref["score"], equal_nan=True)
        pos0 += wi["n_positions"]
    acc.score_condense_all(capi.SCORE_LOGISTIC)
    e3, v3 = acc.download_survivors()
    assert np.array_equal(e3, e1)
    for f in ("cand_index", "record"):
        assert np.array_equal(v3[f], v1[f]), f
    assert np.array_equal(v3["score"], v1["score"], equal_nan=True)
    # forced window starts (ABI 5: the multi-device front end deals region blocks, a window never spans two): with room for everything in one window
    # the batch is cut exactly at the breaks; with one region per window the breaks change nothing; the results are the unbroken batch's
    if len(regions) >= 3:
        acc.set_window_candidates(0)
        acc.set_window_breaks([1, len(regions) - 1])
        grids4, s4, r4 = acc.score_regions(regions, capi.SCORE_LOGISTIC)
        assert acc.window_count() == (3 if len(regions) > 2 else 2)
        assert [acc.window_info(w)["first_region"] for w in range(acc.window_count())] == [0, 1, len(regions) - 1]
        assert np.array_equal(s1, s4, equal_nan=True) and np.array_equal(r1, r4)
        acc.score_condense_all(capi.SCORE_LOGISTIC)
        e4, v4 = acc.download_survivors()
        assert np.array_equal(e4, e1) and np.array_equal(v4["cand_index"], v1["cand_index"])
        with pytest.raises(capi.AccelError):
            acc.set_window_breaks([2, 1])                                          # not ascending
        acc.set_window_breaks([])
        acc.upload(regions)
        assert acc.window_count() == 1
    acc.close()


@pytest.mark.parametrize("name", ["logistic_default_arms", "svr_small", "mixed_12_regions", "hard_logistic", "hard_svr", "hard_saturated_logistic", "hard_ties_svr"])
def test_score_condense_window_leaves_the_survivors_of_the_two_call_route(name, genome):
    """mipgen_accel_score_condense_window (ABI 6: what the silent front end calls per result window) = mipgen_accel_score_window +
    mipgen_accel_replay_condense for a caller that never reads the dense results: emitted counts and survivors - index, record and the score to the bit,
    print-exact re-scores included (tested on the survivors only on this route, on every dense candidate on the other) - are the same, window after
    window, and so is the collapse that follows."""
    meta = H.load_design(name)
    P = H.design_params(meta)
    regions = H.design_regions(meta, _genome_of(meta, genome), P, lrc_fn=po.long_range_content)
    method = capi.SCORE_SVR if meta["method"] == "svr" else capi.SCORE_LOGISTIC
    acc = capi.Accel(P)
    if meta["model"]:
        acc.load_model_file(_model_path(meta))
    grids = acc.upload(regions)
    acc.set_window_candidates(int(max(g.count for g in grids)) + 1)              # one region per window
    acc.upload(regions)
    assert acc.window_count() >= 1
    for w in range(acc.window_count()):
        acc.score_window(w, method)
        acc.replay_condense()
        ea, va, _ = acc.download_replay(window=w)
        acc.collapse()
        ca = acc.download_collapsed(w)
        acc.score_condense_window(w, method)
        eb, vb, _ = acc.download_replay(want_mask=False, window=w)
        acc.collapse()
        cb = acc.download_collapsed(w)
        assert np.array_equal(ea, eb), (name, w)
        for f in ("cand_index", "record"):
            assert np.array_equal(va[f], vb[f]), (name, w, f)
        assert np.array_equal(va["score"].view(np.uint64), vb["score"].view(np.uint64)), (name, w)
        assert np.array_equal(ca, cb), (name, w)
        with pytest.raises(capi.AccelError):
            acc.download_replay(want_mask=True, window=w)                         # the emitted flags are not kept on this route
    with pytest.raises(capi.AccelError):
        acc.score_condense_window(acc.window_count(), method)
    acc.close()


def test_copy_numbers_beyond_16_bits(genome):
    """bwa's X0 count is unbounded (mipgen.cpp:586-587); the 64-bit record saturates its copy fields at 65535.  The condense fold compares
    the true counts (a candidate with copy 70,000 beats one with 80,000 at :1709): the device fetches them from the copy table."""
    Pc = capi.make_params(152, 162, arm_pairs=[(20, 24), (21, 23), (22, 22)], max_arm_copy_product=2_000_000_000)
    rd = capi.build_region(genome, "1", 7000, 7080, Pc, bwa_mode="hashed", label="big")
    rng = np.random.default_rng(21)
    for ln, tab in rd.copy.items():
        hit = rng.random(tab.shape[0]) < 0.5
        tab[hit] = rng.integers(65_000, 90_000, size=int(hit.sum()))
    rd2 = capi.RegionData(rd.c.start_flanked, rd.c.stop_flanked, rd.c.seq_start, rd.seq, copy=rd.copy, unmappable=rd.unmappable,
                          chrom="1", label="big", start=rd.start, stop=rd.stop)
    acc = capi.Accel(Pc)
    grids, scores, records = acc.score_regions([rd2], capi.SCORE_LOGISTIC)
    assert (capi.rec_ext_copy(records) == 65535).any()
    acc.replay_condense()
    emitted, surv, mask = acc.download_replay()
    n_emit, omask = po.replay_region(Pc, rd2, scores, records)
    osurv = po.condense_region(Pc, rd2, scores, records, omask)
    assert np.array_equal(mask, omask)
    assert np.array_equal(surv["cand_index"], osurv["cand_index"])
    assert np.array_equal(surv["record"], osurv["record"])
    acc.close()


def test_long_range_batch(genome):
    """mipgen_accel_long_range_content_batch: one launch, one workgroup per region; bit-exact against the oracle."""
    P = capi.make_params(120, 130)
    acc = capi.Accel(P)
    rng = np.random.default_rng(8)
    seqs, starts, stops = [], [], []
    for _ in range(37):
        a = int(rng.integers(0, len(genome) - 4000)); ln = int(rng.integers(2100, 3900))
        seqs.append(genome[a:a + ln]); starts.append(a + 1001); stops.append(a + ln - 1000)
    got = acc.long_range_content_batch(seqs, starts, stops)
    for i in range(len(seqs)):
        assert np.array_equal(got[i], po.long_range_content(seqs[i], starts[i], stops[i]))
    acc.close()


def test_parameter_sets_outside_the_dense_svr_limits(genome, tmp_path):
    """More than 240 arm pairs, scan sizes below 3 and -capture_increment 1 over a 100-base range are outside the tiled SVR kernel's limits;
    the reference accepts any -arm_lengths / -capture_increment / range (mipgen.cpp:222-261, 427-444).  Logistic designs never were bound by
    them; SVR requests on such a handle take the list route (every dense candidate through k_features_batch + k_svr_gemm) and must give the
    oracle's dense grid: records bit-exact, scores within 1e-5, the replayed enumeration and the condensed survivors identical."""
    from mipgen_amd import synth
    # (a) 255 arm pairs: logistic as before, SVR through the list route
    pairs = [(e, l) for e in range(16, 31) for l in range(14, 31)]          # 255 pairs
    P = capi.make_params(150, 155, arm_pairs=pairs)
    acc = capi.Accel(P)
    rd = capi.build_region(genome, "1", 9000, 9030, P, bwa_mode="hashed", label="many", lrc=np.full(44, 0.02))
    grids, scores, records = acc.score_regions([rd], capi.SCORE_LOGISTIC)
    _, os_, or_ = po.score_region_dense(P, rd, capi.SCORE_LOGISTIC, None)
    assert np.array_equal(records, or_)
    ok, mx = _close(scores, os_)
    assert ok.all(), mx
    mp64 = os.path.join(H.GOLDEN, "models", "svr_syn_64.model")
    acc.load_model_file(mp64)
    om = po.Model(mp64)
    acc.score_window(0, capi.SCORE_SVR)
    s, r = acc.download()
    _, os_, or_ = po.score_region_dense(P, rd, capi.SCORE_SVR, om)
    assert np.array_equal(r, or_)
    ok, mx = _close(s, os_)
    assert ok.all(), mx
    assert np.unique(np.round(s[np.isfinite(s)], 6)).size > 1000             # real kernel values, not one constant
    acc.close()
    # (b) arm sums 30..60 (195 pairs), capture 62-162 in steps of 1, scan sizes from 2: a model whose support vectors come from short captures
    mp = str(tmp_path / "short.model")
    synth.synthetic_svr_model(mp, genome, 48, seed=5, gamma=0.004, capture=(62, 120))
    om = po.Model(mp)
    pairs = synth.arm_pairs_from_sums(list(range(30, 61)))
    assert len(pairs) == 195 and max(e + l for e, l in pairs) == 60
    P = capi.make_params(62, 162, score_method=capi.SCORE_SVR, arm_pairs=pairs, capture_increment=1)
    acc = capi.Accel(P)
    acc.load_model_file(mp)
    rd = capi.build_region(genome, "1", 9000, 9056, P, bwa_mode="hashed", label="wide", lrc=np.linspace(0.01, 0.3, 44))
    grids, scores, records = acc.score_regions([rd], capi.SCORE_SVR)
    og, os_, or_ = po.score_region_dense(P, rd, capi.SCORE_SVR, om)
    assert grids[0].count == og.count and og.n_sizes >= 20
    assert np.array_equal(records, or_)
    ok, mx = _close(scores, os_)
    assert ok.all(), mx
    fin = scores[np.isfinite(scores) & (capi.rec_flags(records) & capi.FLAG_VALID != 0)]
    assert np.unique(np.round(fin, 6)).size > 1000
    acc.replay_condense()
    em, sv, mask = acc.download_replay()
    n_emit, omask = po.replay_region(P, rd, scores, records)
    assert int(em[0]) == n_emit and np.array_equal(mask, omask)
    osurv = po.condense_region(P, rd, scores, records, omask)
    assert np.array_equal(sv["cand_index"], osurv["cand_index"]) and np.array_equal(sv["record"], osurv["record"])
    # the silent path (what the command line runs) takes the same route
    acc.score_condense_all(capi.SCORE_SVR)
    em2, sv2 = acc.download_survivors()
    assert int(em2[0]) == n_emit and np.array_equal(sv2["cand_index"], osurv["cand_index"])
    acc.close()
    # (c) capture sizes of 1,290-1,300 bases: scan sizes beyond the 1,024 bases the list kernels stage at a time (the insert's mers and class switches are
    # counted piece by piece); both scoring methods, the dense grid and the literal single-candidate route
    P = capi.make_params(1290, 1300, score_method=capi.SCORE_SVR)
    acc = capi.Accel(P)
    mpl = str(tmp_path / "long.model")                                      # support vectors drawn from captures of this length: kernel values of O(1)
    synth.synthetic_svr_model(mpl, genome, 48, seed=9, gamma=2e-4, capture=(1285, 1300))
    acc.load_model_file(mpl)
    om = po.Model(mpl)
    rd = capi.build_region(genome, "1", 9000, 9012, P, bwa_mode="hashed", label="long", lrc=np.linspace(0.02, 0.2, 44))
    for method, model in ((capi.SCORE_SVR, om), (capi.SCORE_LOGISTIC, None)):
        grids, scores, records = acc.score_regions([rd], method)
        og, os_, or_ = po.score_region_dense(P, rd, method, model)
        assert grids[0].count == og.count and np.array_equal(records, or_)
        ok, mx = _close(scores, os_)
        assert ok.all(), (method, mx)
        valid = (capi.rec_flags(records) & capi.FLAG_VALID) != 0
        assert valid.sum() > 1000 and np.unique(np.round(scores[valid], 6)).size > 500
        acc.replay_condense()
        em, sv, mask = acc.download_replay()
        n_emit, omask = po.replay_region(P, rd, scores, records)
        assert int(em[0]) == n_emit and np.array_equal(mask, omask)
    g = grids[0]
    pairs = capi.arm_pairs_of(P)
    cands = [(0, g.first_pos + 3, 1300, pairs[0][0], pairs[0][1], 0), (0, g.first_pos + 5, 1295, pairs[-1][0], pairs[-1][1], 1),
             (0, g.first_pos + 1, 1290, pairs[20][0], pairs[20][1], 1)]
    lrc = np.array(rd.c.long_range_content[:])
    for method, model in ((capi.SCORE_SVR, om), (capi.SCORE_LOGISTIC, None)):
        s_lit, r_lit, _, ints = acc.score_candidates(cands, method, want_ints=True)
        for i, c in enumerate(cands):
            sk, d = po.design(P, rd, c)
            assert not sk and d.scan_size == c[2] - c[3] - c[4] > 1024
            want_s, _, want_ints = po.score_designed(d, method, lrc, model)
            assert (capi.rec_flags(r_lit[i:i + 1]) & capi.FLAG_VALID).all()
            for f in ("ins_a", "ins_c", "ins_g", "ins_t", "run_count", "junction", "scan_size"):
                assert getattr(ints[i], f) == getattr(want_ints, f), (i, f)
            ok, mx = _close(s_lit[i:i + 1], [want_s])
            assert ok.all(), (method, i, mx)
    acc.close()


@pytest.mark.parametrize("n_sv", [1, 2, 4, 7])
def test_tiny_models(genome, tmp_path, n_sv):
    """One to a few support vectors: a single (partial) SV group, two groups, and the prologue / loop boundaries of the kernel."""
    from mipgen_amd import workloads
    mp = workloads.svr_model_path(str(tmp_path), genome, n_sv, seed=11)
    P = capi.make_params(152, 162, score_method=capi.SCORE_SVR)
    acc = capi.Accel(P)
    acc.load_model_file(mp)
    om = po.Model(mp)
    rd = capi.build_region(genome, "1", 9000, 9040, P, bwa_mode="hashed", label="tiny", lrc=np.full(44, 0.02))
    grids, scores, records = acc.score_regions([rd], capi.SCORE_SVR)
    og, os_, or_ = po.score_region_dense(P, rd, capi.SCORE_SVR, om)
    assert (records[:og.count] == or_).all()
    ok, mx = _close(scores[:og.count], os_)
    assert ok.all(), (n_sv, mx)
    acc.close()


def test_survivor_array_as_torch_tensor(tmp_path):
    """The multi-GPU gather sends the library's survivor array straight from HBM: mipgen_accel_survivors_device_ptr wrapped as a torch
    tensor (no copy) holds the same bytes mipgen_accel_download_survivors returns.  Run as bench.py runs: torch first, then the library
    (a fresh process; this one has had the HIP runtime initialised by the library alone)."""
    import subprocess
    import sys
    script = tmp_path / "view.py"
    script.write_text(f"""
import sys
sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})
import numpy as np, torch
torch.cuda.set_device(0)
from mipgen_amd import capi
from tests import helpers as H
meta = H.load_design("logistic_default_arms")
P = H.design_params(meta)
regions = H.design_regions(meta, H.golden_genome(), P)
acc = capi.Accel(P, stream=torch.cuda.current_stream().cuda_stream)
acc.upload(regions)
acc.score_condense_all(capi.SCORE_LOGISTIC)
emitted, surv = acc.download_survivors()
ptr, n = acc.survivors_device_ptr()
assert n == surv.shape[0] > 0
class View:
    __cuda_array_interface__ = {{"shape": (n * 24,), "typestr": "|u1", "data": (ptr, False), "version": 2}}
t = torch.as_tensor(View(), device="cuda:0")
torch.cuda.synchronize()
got = t.cpu().numpy().view(capi.SURVIVOR_DTYPE)
assert np.array_equal(got["cand_index"], surv["cand_index"]) and np.array_equal(got["record"], surv["record"])
acc.close()
print("view ok")
""")
    p = subprocess.run([sys.executable, str(script)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert p.returncode == 0 and b"view ok" in p.stdout, p.stdout.decode()[-2000:]


def test_model_loader_paths(genome, tmp_path):
    """svm_load_model's grammar (svm.cpp:2779-2962) through mipgen_accel_load_model_file: non-RBF / non-SVR models, unknown header
    keys and truncated SV lists are errors with a message; nu_svr loads; libsvm indices beyond 192 (x has none: they contribute
    sv^2 to every distance, svm.cpp:359-363) load and score as the oracle's merge walk does."""
    base = open(os.path.join(H.GOLDEN, "models", "svr_syn_64.model")).read()
    P = capi.make_params(130, 140, score_method=capi.SCORE_SVR, arm_pairs=synth.arm_pairs_from_sums([44, 45]))
    acc = capi.Accel(P)

    def write(name, text):
        p = tmp_path / name
        p.write_text(text)
        return str(p)
    with pytest.raises(capi.AccelError, match="rbf only"):
        acc.load_model_file(write("lin.model", base.replace("kernel_type rbf", "kernel_type linear")))
    with pytest.raises(capi.AccelError, match="SVR only"):
        acc.load_model_file(write("csvc.model", base.replace("svm_type epsilon_svr", "svm_type c_svc")))
    with pytest.raises(capi.AccelError, match="unknown text"):
        acc.load_model_file(write("junk.model", base.replace("nr_class 2", "nr_klass 2")))
    lines = base.split("\n")
    with pytest.raises(capi.AccelError, match="SV lines"):
        acc.load_model_file(write("short.model", "\n".join(lines[:-6]) + "\n"))
    with pytest.raises(capi.AccelError, match="unknown text"):                # the SV line is missing: the first coefficient is read as a header key (svm.cpp:2893)
        acc.load_model_file(write("nosv.model", "\n".join(l for l in lines if l.strip() != "SV")[:400]))
    with pytest.raises(capi.AccelError, match="malformed"):
        acc.load_model_file(write("badgamma.model", base.replace("gamma 0.0104167", "gamma abc").replace("gamma 0.010416", "gamma abc")))
    acc.load_model_file(write("nu.model", base.replace("svm_type epsilon_svr", "svm_type nu_svr")))
    assert acc.model_info()[0] == 64
    # indices > 192 on a third of the SV lines
    i_sv = lines.index("SV") + 1
    ext = lines[:i_sv] + [l.rstrip() + (f" 193:{0.25 + 0.01 * k:.4g} 260:{0.5 - 0.003 * k:.4g} " if k % 3 == 0 and l.strip() else "") for k, l in enumerate(lines[i_sv:])]
    mp = write("extra.model", "\n".join(ext))
    acc.load_model_file(mp)
    om = po.Model(mp)
    rd = capi.build_region(genome, "1", 5000, 5060, P, bwa_mode="hashed", label="x", lrc=np.linspace(0.02, 0.3, 44))
    grids, scores, records = acc.score_regions([rd], capi.SCORE_SVR)
    _, os_, or_ = po.score_region_dense(P, rd, capi.SCORE_SVR, om)
    ok, mx = _close(scores, os_)
    assert ok.all(), mx
    base_acc = capi.Accel(P)
    base_acc.load_model_file(os.path.join(H.GOLDEN, "models", "svr_syn_64.model"))
    _, s0, _ = base_acc.score_regions([rd], capi.SCORE_SVR)
    valid = (capi.rec_flags(records) & capi.FLAG_VALID) != 0
    assert np.nanmax(np.abs(s0[valid] - scores[valid])) > 1e-6            # the extra indices do change the scores
    base_acc.close()
    acc.close()


def test_model_written_by_libsvm_itself_through_the_accelerator(genome):
    """The model the reference's own libsvm trained and wrote (tests/golden/models/svr_libsvm_trained.model: svm_train + svm_save_model, svm.cpp:2095,
    2644-2757 - genuine output, 227 sparse SV lines at %.8g) through mipgen_accel_load_model_file: header as svm_load_model reads it, and the dense
    grid of a region - tiled kernel and literal per-candidate kernel - within 1e-5 of the oracle, whose predictions are bit-exact against the
    reference's svm_predict on that file (tests/test_oracle_golden.py)."""
    z = np.load(os.path.join(H.GOLDEN, "libsvm_trained.npz"))
    mp = os.path.join(H.GOLDEN, "models", "svr_libsvm_trained.model")
    P = capi.make_params(135, 150, score_method=capi.SCORE_SVR, arm_pairs=synth.arm_pairs_from_sums([43, 44, 45]))
    acc = capi.Accel(P)
    acc.load_model_file(mp)
    n_sv, gamma, rho = acc.model_info()
    assert n_sv == int(z["n_sv"][0]) and gamma == float(z["gamma"][0]) and rho == float(z["rho"][0])
    om = po.Model(mp)
    rd = capi.build_region(genome, "1", 7000, 7080, P, bwa_mode="hashed", label="x", lrc=np.linspace(0.02, 0.3, 44))
    grids, scores, records = acc.score_regions([rd], capi.SCORE_SVR)
    _, os_, or_ = po.score_region_dense(P, rd, capi.SCORE_SVR, om)
    assert np.array_equal(records, or_)
    ok, mx = _close(scores, os_)
    assert ok.all(), mx
    # the literal per-candidate kernel (reference operation order) on a sample of the grid
    g = grids[0]
    rng = np.random.default_rng(5)
    A = P.n_arm_pairs
    cands, idxs = [], []
    for idx in rng.choice(g.count, size=200, replace=False):
        a = int(idx % A); row = int(idx // A); st = row & 1; rest = row >> 1
        ki, pi = rest % g.n_sizes, rest // g.n_sizes
        if not (int(capi.rec_flags(records[idx:idx + 1])[0]) & capi.FLAG_VALID):
            continue
        cands.append((0, g.first_pos + pi, P.max_capture_size - (g.first_size_index + ki) * P.capture_increment, P.arm_ext[a], P.arm_lig[a], int(st)))
        idxs.append(int(idx))
    lit = acc.score_candidates(cands, capi.SCORE_SVR)[0]
    ok, mx = _close(lit, os_[np.array(idxs)])
    assert ok.all(), mx
    acc.close()


def _collapse_py(P, g, surv, target, max_product, thr):
    """collapse_mips (mipgen.cpp:1616-1649) restated as a plain loop over the survivors in scan-start order (test-side checker)."""
    A = P.n_arm_pairs
    max_scan = P.max_capture_size - g.first_size_index * P.capture_increment - min(P.arm_ext[i] + P.arm_lig[i] for i in range(A))
    n_base = g.n_pos + max_scan - 1 if g.n_pos > 0 and g.n_sizes > 0 else 0
    best = -np.ones((n_base, 2), dtype=np.int32)
    state = {}
    for pi in range(g.n_pos):
        for s in range(2):
            m = surv[2 * pi + s]
            if m["cand_index"] < 0:
                continue
            rel = int(m["cand_index"]) - g.offset - pi * g.n_sizes * A * 2
            a, ki = rel % A, rel // (2 * A)
            e, l = P.arm_ext[a], P.arm_lig[a]
            ss = P.max_capture_size - (g.first_size_index + ki) * P.capture_increment - e - l
            r = int(m["record"])
            ec, lc = r & 0xFFFF, (r >> 16) & 0xFFFF
            if ec * lc > max_product or ec > target or lc > target:
                continue
            if ((r >> 32) & 0xFF) / (l + e) > thr:
                continue
            snp, sc = (r >> 40) & 0xFF, float(m["score"])
            for j in range(pi, pi + ss):
                cur = state.get((j, s))
                if cur is None or snp < cur[0] or (sc > cur[1] and snp == cur[0]):
                    state[(j, s)] = (snp, sc)
                    best[j, s] = pi
    return best


@pytest.mark.parametrize("name", ["logistic_snp_trf", "mixed_small", "svr_small", "hard_mixed"])
def test_collapse_on_device(name, genome):
    """mipgen_accel_collapse: per base and strand the survivor the reference's collapse fold keeps (SNP count first, then strictly higher
    score, first come first kept), incl. the copy / masked-arm filters - against a plain restatement of the fold; window by window and
    through the fused silent path."""
    meta = H.load_design(name)
    P = H.design_params(meta)
    regions = H.design_regions(meta, _genome_of(meta, genome), P, lrc_fn=po.long_range_content)
    method = capi.SCORE_SVR if meta["method"] == "svr" else capi.SCORE_LOGISTIC
    acc = capi.Accel(P)
    if meta["model"]:
        acc.load_model_file(_model_path(meta))
    acc.set_window_candidates(int(max(po.grid(P, r).count for r in regions)) + 1)
    grids = acc.upload(regions)
    per_window = []
    for w in range(acc.window_count()):
        acc.score_window(w, method)
        acc.replay_condense()
        acc.collapse()
        e, surv, _ = acc.download_replay(want_mask=False, window=w)
        col = acc.download_collapsed(w)
        wi = acc.window_info(w)
        off = 0
        pos = 0
        for ri in range(wi["first_region"], wi["first_region"] + wi["n_regions"]):
            g = grids[ri]
            fe, nb = acc.region_bases(ri)
            exp = _collapse_py(P, g, surv[2 * pos:2 * (pos + g.n_pos)], P.target_arm_copy, P.max_arm_copy_product, P.masked_arm_threshold)
            assert exp.shape[0] == nb
            assert np.array_equal(col[off:off + 2 * nb].reshape(nb, 2), exp), (name, ri)
            assert (exp >= 0).any()
            off += 2 * nb; pos += g.n_pos
        per_window.append(col)
    acc.score_condense_all(method)
    assert np.array_equal(acc.download_collapsed(-1), np.concatenate(per_window))
    acc.close()


@pytest.mark.parametrize("capture", [(300, 320), (1500, 1510)])
def test_collapse_wide_scan_targets(capture):
    """The collapse kernel parks the survivors a 128-base tile can see in LDS: wide capture sizes (scan targets of ~270 and ~1,470 bases: the
    second needs more than the default 48 KB of dynamic LDS) against the plain restatement, over several tiles of a long region."""
    g = synth.random_genome(12_000, 91, n_run_frac=0.001, n_run_len=6)
    P = capi.make_params(capture[0], capture[1], score_method=capi.SCORE_LOGISTIC, capture_increment=10, arm_pairs=[(20, 20), (20, 24), (22, 22), (24, 20)])
    from mipgen_amd import workloads
    regions = [workloads.fast_region(g, synth.Interval("1", 4000, 4700, "a"), P), workloads.fast_region(g, synth.Interval("1", 8000, 8150, "b"), P)]
    acc = capi.Accel(P)
    grids = acc.upload(regions)
    acc.score_condense_all(capi.SCORE_LOGISTIC)
    e, surv, _ = acc.download_replay(want_mask=False)
    col = acc.download_collapsed(-1)
    off = pos = 0
    for ri, gr in enumerate(grids):
        fe, nb = acc.region_bases(ri)
        exp = _collapse_py(P, gr, surv[2 * pos:2 * (pos + gr.n_pos)], P.target_arm_copy, P.max_arm_copy_product, P.masked_arm_threshold)
        assert exp.shape[0] == nb
        assert np.array_equal(col[off:off + 2 * nb].reshape(nb, 2), exp), (capture, ri)
        assert (exp >= 0).any()
        off += 2 * nb; pos += gr.n_pos
    acc.close()


def test_batched_candidate_rescoring(genome):
    """Lists of >= 256 candidates take the list path (k_features_batch: a wavefront per candidate, then k_svr_gemm: distances through the
    FP64 matrix cores): same features / records / scores as the one-workgroup-per-candidate kernel and as the oracle, incl. guard /
    zero-copy / invalid candidates."""
    meta = H.load_design("mixed_small")
    P = H.design_params(meta)
    regions = H.design_regions(meta, genome, P, lrc_fn=po.long_range_content)
    mp = os.path.join(H.GOLDEN, "models", "svr_syn_200.model")
    acc = capi.Accel(P)
    acc.load_model_file(mp)
    om = po.Model(mp)
    grids = acc.upload(regions)
    rng = np.random.default_rng(6)
    A = P.n_arm_pairs
    cands = []
    for ri, g in enumerate(grids):
        for idx in rng.choice(g.count, size=700, replace=False):
            a = idx % A; row = idx // A; strand = row & 1; rest = row >> 1
            ki, pi = rest % g.n_sizes, rest // g.n_sizes
            cands.append((ri, g.first_pos + int(pi), P.max_capture_size - (g.first_size_index + int(ki)) * P.capture_increment, P.arm_ext[int(a)], P.arm_lig[int(a)], int(strand)))
    cands.append((0, 3, 130, 20, 22, 0))                                    # fails the bounds skips
    batched, rec_b, feat_b, _ = acc.score_candidates(cands, capi.SCORE_SVR, want_features=True)
    parts = [acc.score_candidates(cands[i:i + 100], capi.SCORE_SVR, want_features=True) for i in range(0, len(cands), 100)]
    single = np.concatenate([q[0] for q in parts])
    assert np.nanmax(np.abs(batched - single)) < 1e-10
    # the list path's feature / record kernel (one wavefront per candidate, histogram mer counts) against the per-candidate kernel: bit for bit
    assert np.array_equal(rec_b, np.concatenate([q[1] for q in parts]))
    assert np.array_equal(feat_b, np.concatenate([q[2] for q in parts]), equal_nan=True)
    assert batched[-1] == 0.0 and (capi.rec_flags(rec_b[-1:]) & capi.FLAG_VALID)[0] == 0
    n_guard = 0
    for k in rng.choice(len(cands) - 1, size=250, replace=False):
        c = cands[int(k)]
        sk, d = po.design(P, regions[c[0]], (0,) + c[1:])
        so, _, _ = po.score_designed(d, capi.SCORE_SVR, np.array(regions[c[0]].c.long_range_content[:]), om)
        assert abs(batched[int(k)] - so) <= 1e-5 or (np.isnan(so) and np.isnan(batched[int(k)])), (c, batched[int(k)], so)
        n_guard += int((capi.rec_flags(rec_b[int(k):int(k) + 1]) & capi.FLAG_GUARD)[0] != 0)
    acc.close()


@pytest.mark.parametrize("name", ["logistic_snp_trf", "svr_small", "mixed_small"])
def test_all_mips_records_formatted_on_device(name, genome):
    """mipgen_accel_format_all_mips: the window's constructed candidates as print_details writes them (mipgen.cpp:765-794), numbered in
    generation order - byte for byte the reference binary's all_mips file (less its header; the reference's uninitialised masking byte of
    mapping-failed records normalised, tests/helpers.py), also when the batch is cut into windows numbered on from each other."""
    meta = H.load_design(name)
    P = H.design_params(meta)
    regions = H.design_regions(meta, genome, P, lrc_fn=po.long_range_content)
    method = capi.SCORE_SVR if meta["method"] == "svr" else capi.SCORE_LOGISTIC
    ref = [H.normalise_flags(l) for l in H.ref_lines(meta, "all_mips")[1:]]
    for cap in (0, 1):
        acc = capi.Accel(P)
        if meta["model"]:
            acc.load_model_file(_model_path(meta))
        if cap:
            acc.set_window_candidates(int(max(po.grid(P, r).count for r in regions)) + 1)
        acc.upload(regions)
        text, first = b"", 0
        for w in range(acc.window_count()):
            wi = acc.window_info(w)
            acc.score_window(w, method)
            acc.replay_condense()
            names = [("1", regions[i].label, regions[i].start - 1, regions[i].stop) for i in range(wi["first_region"], wi["first_region"] + wi["n_regions"])]
            t, n = acc.format_all_mips(names, H.middle_of(meta["tags"]), first)
            text += t; first += n
        acc.close()
        got = text.split(b"\n")
        assert got[-1] == b"" and len(got) - 1 == len(ref) == first
        bad = [i for i, (a, b) in enumerate(zip(got, ref)) if a != b]
        assert not bad, (name, cap, bad[0], got[bad[0]][:160], ref[bad[0]][:160])


def test_list_scorer_special_values(genome):
    """The list path (k_features_batch + k_svr_gemm) on the candidates whose feature vectors are not ordinary numbers: an N in an arm (all-zero
    vector, SVMipv4.cpp:63-68), a copy number of 0 (log10 = -inf: every kernel value 0, score -rho), a negative copy number (log10 = NaN: the
    score is NaN) and, for contrast, copy numbers above 100 (feature 2.0).  Scores against the oracle's own libsvm arithmetic; features and
    records bit-identical to the per-candidate kernel."""
    g = bytearray(genome)
    g[9_100:9_104] = b"NNNN"
    g = bytes(g)
    P = capi.make_params(150, 160, score_method=capi.SCORE_SVR)
    mp = os.path.join(H.GOLDEN, "models", "svr_syn_200.model")
    om = po.Model(mp)
    rd0 = capi.build_region(g, "1", 9_000, 9_200, P, bwa_mode="unique", label="sp", lrc=np.full(44, 0.02))
    lengths = sorted({e for e, _ in capi.arm_pairs_of(P)} | {l for _, l in capi.arm_pairs_of(P)})
    rng = np.random.default_rng(17)
    copy = {}
    for ln in lengths:
        c = np.ones(rd0.c.seq_len, dtype=np.int32)
        u = rng.random(rd0.c.seq_len)
        c[u < 0.05] = 0
        c[(u >= 0.05) & (u < 0.08)] = -3
        c[(u >= 0.08) & (u < 0.12)] = 250
        copy[ln] = c
    rd = capi.RegionData(rd0.c.start_flanked, rd0.c.stop_flanked, rd0.c.seq_start, rd0.seq, masked=rd0.masked if rd0.masked is not None else rd0.seq,
                         copy=copy, lrc=[0.02] * 44, chrom="1", label="sp",
                         start=rd0.start, stop=rd0.stop)
    acc = capi.Accel(P)
    acc.load_model_file(mp)
    grids = acc.upload([rd])
    gr = grids[0]
    A = P.n_arm_pairs
    cands = []
    for idx in rng.choice(gr.count, size=3000, replace=False):
        a = idx % A; row = idx // A; strand = row & 1; rest = row >> 1
        ki, pi = rest % gr.n_sizes, rest // gr.n_sizes
        cands.append((0, gr.first_pos + int(pi), P.max_capture_size - (gr.first_size_index + int(ki)) * P.capture_increment, P.arm_ext[int(a)], P.arm_lig[int(a)], int(strand)))
    sc, rec, feat, _ = acc.score_candidates(cands, capi.SCORE_SVR, want_features=True)
    parts = [acc.score_candidates(cands[i:i + 200], capi.SCORE_SVR, want_features=True) for i in range(0, len(cands), 200)]
    assert np.array_equal(rec, np.concatenate([q[1] for q in parts]))
    assert np.array_equal(feat, np.concatenate([q[2] for q in parts]), equal_nan=True)
    one = np.concatenate([q[0] for q in parts])
    assert np.array_equal(np.isnan(sc), np.isnan(one))
    assert np.nanmax(np.abs(sc - one)) < 1e-10
    flags = capi.rec_flags(rec)
    valid = (flags & capi.FLAG_VALID) != 0
    guard = valid & ((flags & capi.FLAG_GUARD) != 0)
    assert guard.sum() > 20 and np.isnan(sc).sum() > 20 and np.isinf(feat).any() and (feat[:, 190:192] == 2.0).any()
    rho = acc.model_info()[2]
    zero = valid & ~guard & ~np.isnan(sc) & (np.isinf(feat[:, 190]) | np.isinf(feat[:, 191]))
    assert zero.sum() > 20 and np.all(sc[zero] == -rho)
    lrc = np.array(rd.c.long_range_content[:])
    pick = np.concatenate([np.nonzero(guard)[0][:30], np.nonzero(np.isnan(sc))[0][:30], np.nonzero(zero)[0][:30], rng.choice(len(cands), 150, replace=False)])
    for k in pick:
        c = cands[int(k)]
        sk, d = po.design(P, rd, (0,) + c[1:])
        if sk:
            assert sc[int(k)] == 0.0
            continue
        so, _, _ = po.score_designed(d, capi.SCORE_SVR, lrc, om)
        assert abs(sc[int(k)] - so) <= 1e-5 or (np.isnan(so) and np.isnan(sc[int(k)])), (c, sc[int(k)], so)
    acc.close()


@pytest.mark.parametrize("method", [capi.SCORE_LOGISTIC, capi.SCORE_SVR])
def test_arm_sum_keys_with_empty_lists(genome, method):
    """`-arm_length_sums 30,41,43,62` with the default minimum arm lengths: the lists of 30 and 62 are empty but their keys stay in the reference's map
    (mipgen.cpp:245-258), so the first scan position is computed from 62 (:421) and no list is exempt from the switch-off of :434 (the exempt one is the
    list of the smallest KEY).  The boundary carries the two keys (mipgen_params.arm_sum_key_max / _min); grid, replay mask, emitted count and survivors
    against the oracle, with thresholds low enough for the switch-off to happen; the same pairs WITHOUT the keys must give a different grid and mask."""
    pairs = synth_pairs = __import__("mipgen_amd.synth", fromlist=["x"]).arm_pairs_from_sums([41, 43])
    kw = dict(score_method=method, arm_pairs=pairs, logistic_optimal=0.9, svr_optimal=1.5)
    mp64 = os.path.join(H.GOLDEN, "models", "svr_syn_64.model")
    om = po.Model(mp64) if method == capi.SCORE_SVR else None
    got = {}
    for keys in ((62, 30), None):
        P = capi.make_params(150, 160, arm_sum_keys=keys, **kw)
        acc = capi.Accel(P)
        if om is not None:
            acc.load_model_file(mp64)
        rd = capi.build_region(genome, "1", 9000, 9120, P, bwa_mode="hashed", label="keys", lrc=np.full(44, 0.03))
        grids, scores, records = acc.score_regions([rd], method)
        og, os_, or_ = po.score_region_dense(P, rd, method, om)
        g = grids[0]
        assert (g.first_pos, g.n_pos, g.count) == (og.first_pos, og.n_pos, og.count) and np.array_equal(records, or_)
        acc.replay_condense()
        em, sv, mask = acc.download_replay()
        n_emit, omask = po.replay_region(P, rd, scores, records)
        assert int(em[0]) == n_emit and np.array_equal(mask, omask)
        osurv = po.condense_region(P, rd, scores, records, omask)
        assert np.array_equal(sv["cand_index"], osurv["cand_index"])
        acc.score_condense_all(method)
        em2, sv2 = acc.download_survivors()
        assert int(em2[0]) == n_emit and np.array_equal(sv2["cand_index"], osurv["cand_index"])
        got[keys] = (g.first_pos, g.n_pos, n_emit)
        acc.close()
    assert got[(62, 30)][0] == got[None][0] + 19 and got[(62, 30)][1] == got[None][1] - 19            # 62 - 43 positions fewer
    assert got[(62, 30)][2] != got[None][2]
    # keys inside the pairs' own sums are refused
    P = capi.make_params(150, 160, arm_sum_keys=(42, 0), **kw)
    with pytest.raises(RuntimeError, match="arm_sum_key"):
        capi.Accel(P)

"""Seeded randomized parity: the dense GPU path against the oracle over parameter sets the fixed tests do not reach
(capture increments other than 5, narrow and wide capture ranges, arm-pair subsets in random order, short and long regions,
N runs, hashed copy tables; every second configuration on the hard genome: ambiguity codes, homopolymers, microsatellites).  Records bit-exact on every candidate; scores within 1e-5 on a sample (SVR) / everywhere (logistic).
"""
import os

import numpy as np
import pytest

from mipgen_amd import capi
from oracle import pyoracle as po
from tests import helpers as H

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _configs():
    rng = np.random.default_rng(20240607)
    grid = [(e, l) for e in range(16, 31) for l in range(18, 31) if 38 <= e + l <= 48]
    out = []
    for i in range(int(os.environ.get("MIPGEN_FUZZ_N", "24"))):          # a longer soak: MIPGEN_FUZZ_N=200 python -m pytest tests/test_gpu_fuzz.py
        inc = int(rng.choice([1, 2, 3, 5, 7, 10]))
        lo = int(rng.integers(110, 170))
        hi = lo + inc * int(rng.integers(0, 12))
        n_pairs = int(rng.integers(1, 40))
        idx = rng.permutation(len(grid))[:n_pairs]
        pairs = [grid[j] for j in (sorted(idx, key=lambda j: (-(grid[j][0] + grid[j][1]), grid[j][0])) if i % 2 == 0 else idx)]
        length = int(rng.choice([1, 7, 40, 120, 260]))
        start = int(rng.integers(2000, 17000))
        out.append((i, lo, hi, inc, pairs, start, length))
    return out


@pytest.mark.parametrize("cfg", _configs(), ids=lambda c: f"cfg{c[0]}_C{c[1]}-{c[2]}x{c[3]}_A{len(c[4])}_L{c[6]}")
def test_random_configuration(cfg):
    i, lo, hi, inc, pairs, start, length = cfg
    # every second configuration on the hard genome: its regions [2,000, 17,260) hold ambiguity codes, upper-cased soft-masked stretches,
    # homopolymers and microsatellites (mipgen_amd/synth.py: hard_genome)
    genome = bytearray(H.golden_genome() if i % 2 == 0 else H.golden_genome("genome4_chr4.fa.gz").upper())
    rng = np.random.default_rng(1000 + i)
    if i % 3 == 0:                                   # an N run inside the region's reach (guard / masked-N paths)
        p0 = start + int(rng.integers(-100, 100))
        genome[p0:p0 + 6] = b"NNNNNN"
    genome = bytes(genome)
    mp = os.path.join(H.GOLDEN, "models", "svr_syn_64.model")
    om = po.Model(mp)
    for method in (capi.SCORE_LOGISTIC, capi.SCORE_SVR):
        P = capi.make_params(lo, hi, score_method=method, capture_increment=inc, arm_pairs=pairs)
        acc = capi.Accel(P)
        if method == capi.SCORE_SVR:
            acc.load_model_file(mp)
        acc.set_logistic_subruns(i % 5)                  # 0 = automatic (1 at this size); 1..4 position sub-runs per tile: the sliding-table paths
        # SNP tables (records' SNP counts and flags: classes 1 = usable allele pair, 2 = anything else, multi-base records), TRF-style masks (masked-base
        # counts, the masking flag), the three copy-number modes of the bwa stand-in (hashed: copies up to 500 and unmappable windows; blocks: dead zones)
        snp_tab = None
        if i % 4 in (1, 2):
            r3 = np.random.default_rng(3000 + i)
            snp_tab = {}
            for pos in r3.integers(max(1, start - 300), start + length + 300, size=int(r3.integers(1, 40))):
                g = chr(genome[int(pos) - 1])
                u = r3.random()
                snp_tab[int(pos)] = (g + "ACGT"[int(r3.integers(0, 4))]) if u < 0.7 else ("ACGT"[int(r3.integers(0, 4))] + "T" if u < 0.85 else g + "AC")
        rd = capi.build_region(genome, "1", start, start + length, P, bwa_mode=("unique", "hashed", "blocks")[i % 3], label=f"f{i}",
                               lrc=np.full(44, 0.01 * (i + 1)), snp_tab=snp_tab, mask_record=i if i % 5 in (2, 3) else None, flank=(0, 0, 7)[i % 3])
        grids, scores, records = acc.score_regions([rd], method)
        g = grids[0]
        og, os_, or_ = po.score_region_dense(P, rd, method, om if method == capi.SCORE_SVR else None) if g.count <= 400000 else (None, None, None)
        if og is not None:
            assert (g.first_pos, g.n_pos, g.first_size_index, g.n_sizes, g.count) == (og.first_pos, og.n_pos, og.first_size_index, og.n_sizes, og.count)
            bad = np.nonzero(records[:g.count] != or_)[0]
            assert bad.size == 0, (cfg[:4], "record", int(bad[0]))
            a, b = np.asarray(scores[:g.count]), np.asarray(os_)
            both_nan = np.isnan(a) & np.isnan(b)
            with np.errstate(invalid="ignore"):
                d = np.where(both_nan | (np.isinf(a) & (a == b)), 0.0, np.abs(a - b))
            assert np.nanmax(d) <= TOL and not np.isnan(d).any(), (cfg[:4], method, int(np.nanargmax(d)), float(np.nanmax(d)))
            # the replay of the early exits and the condense fold on the same arrays (SNP counts, masked bases and copy numbers steer the fold)
            acc.replay_condense()
            em, sv, mask = acc.download_replay()
            n_emit, omask = po.replay_region(P, rd, scores, records)
            assert int(em[0]) == n_emit and np.array_equal(mask, omask), (cfg[:4], method, "replay")
            osurv = po.condense_region(P, rd, scores, records, omask)
            assert np.array_equal(sv["cand_index"], osurv["cand_index"]) and np.array_equal(sv["record"], osurv["record"]), (cfg[:4], method, "condense")
            # the one-call window route of a silent front end (ABI 6): the same survivors, scores to the bit
            acc.score_condense_window(0, method)
            em2, sv2, _ = acc.download_replay(want_mask=False)
            assert int(em2[0]) == n_emit and all(np.array_equal(sv2[f], sv[f]) for f in ("cand_index", "record")), (cfg[:4], method, "condense_window")
            assert np.array_equal(sv2["score"].view(np.uint64), sv["score"].view(np.uint64)), (cfg[:4], method, "condense_window scores")
        else:                                        # large grids: sampled candidates through the per-candidate oracle
            valid = np.nonzero((capi.rec_flags(records) & capi.FLAG_VALID) != 0)[0]
            A = P.n_arm_pairs
            for idx in rng.choice(valid, size=min(300, len(valid)), replace=False):
                a = idx % A; row = idx // A; strand = row & 1; rest = row >> 1
                ki, pi = rest % g.n_sizes, rest // g.n_sizes
                cand = (0, g.first_pos + pi, P.max_capture_size - (g.first_size_index + ki) * P.capture_increment, P.arm_ext[a], P.arm_lig[a], int(strand))
                sk, dsg = po.design(P, rd, cand)
                so, _, _ = po.score_designed(dsg, method, np.array(rd.c.long_range_content[:]), om if method == capi.SCORE_SVR else None)
                assert abs(scores[idx] - so) <= TOL or (np.isnan(scores[idx]) and np.isnan(so)), (cfg[:4], cand, scores[idx], so)
        acc.close()


def _list_route_configs():
    """Parameter sets on and across the limits of the tiled SVR kernel (scan sizes 1..5, 230..256 arm pairs, increments 1..3): the same handle
    takes the tiled kernel or the list route (accel_score.hip: svr_window_via_list) depending on which side of a limit it falls."""
    rng = np.random.default_rng(20260101)
    out = []
    for i in range(int(os.environ.get("MIPGEN_FUZZ_LIST_N", "6"))):
        if i % 2 == 0:                                # short captures: the smallest scan size is 1..5
            sums = sorted(rng.choice(np.arange(36, 47), size=int(rng.integers(1, 4)), replace=False).tolist(), reverse=True)
            pairs = [(e, s - e) for s in sums for e in range(16, 31) if 18 <= s - e <= 30][:int(rng.integers(3, 30))]
            lo = max(e + l for e, l in pairs) + int(rng.integers(1, 6))
            hi = lo + int(rng.integers(0, 9)) * (inc := int(rng.choice([1, 2, 3])))
        else:                                         # many pairs: 230..256 of them, around the tiled kernel's 240
            n = int(rng.integers(230, 257))
            grid = [(e, l) for e in range(14, 31) for l in range(14, 31)]
            idx = sorted(rng.permutation(len(grid))[:n].tolist(), key=lambda j: (-(grid[j][0] + grid[j][1]), grid[j][0]))
            pairs = [grid[j] for j in idx]
            inc = int(rng.choice([1, 5]))
            lo = int(rng.integers(100, 140)); hi = lo + inc * int(rng.integers(0, 3))
        out.append((i, lo, hi, inc, pairs, int(rng.integers(3000, 16000)), int(rng.choice([1, 9, 30]))))
    return out


@pytest.mark.parametrize("cfg", _list_route_configs(), ids=lambda c: f"list{c[0]}_C{c[1]}-{c[2]}x{c[3]}_A{len(c[4])}_L{c[6]}")
def test_random_configuration_around_the_tiled_kernels_limits(cfg):
    i, lo, hi, inc, pairs, start, length = cfg
    genome = H.golden_genome()
    mp = os.path.join(H.GOLDEN, "models", "svr_syn_short_48.model" if i % 2 == 0 else "svr_syn_64.model")
    om = po.Model(mp)
    P = capi.make_params(lo, hi, score_method=capi.SCORE_SVR, capture_increment=inc, arm_pairs=pairs)
    acc = capi.Accel(P)
    acc.load_model_file(mp)
    rd = capi.build_region(genome, "1", start, start + length, P, bwa_mode="hashed", label=f"l{i}", lrc=np.full(44, 0.02 * (i + 1)))
    grids, scores, records = acc.score_regions([rd], capi.SCORE_SVR)
    g = grids[0]
    assert g.count <= 1_500_000
    og, os_, or_ = po.score_region_dense(P, rd, capi.SCORE_SVR, om)
    assert (g.first_pos, g.n_pos, g.first_size_index, g.n_sizes, g.count) == (og.first_pos, og.n_pos, og.first_size_index, og.n_sizes, og.count)
    assert np.array_equal(records[:g.count], or_), cfg[:4]
    a, b = np.asarray(scores[:g.count]), np.asarray(os_)
    both_nan = np.isnan(a) & np.isnan(b)
    with np.errstate(invalid="ignore"):
        d = np.where(both_nan | (np.isinf(a) & (a == b)), 0.0, np.abs(a - b))
    assert np.nanmax(d) <= TOL and not np.isnan(d).any(), (cfg[:4], int(np.nanargmax(d)), float(np.nanmax(d)))
    acc.replay_condense()
    emitted, surv, mask = acc.download_replay()
    n_emit, omask = po.replay_region(P, rd, scores, records)
    assert emitted[0] == n_emit and np.array_equal(mask, omask), cfg[:4]
    osurv = po.condense_region(P, rd, scores, records, omask)
    assert np.array_equal(surv["cand_index"], osurv["cand_index"]) and np.array_equal(surv["record"], osurv["record"])
    acc.close()


def _pair_lists():
    """Arm-pair lists of many shapes: one pair, one list, lists of unequal length, exactly 63 / 64 pairs (the single-wavefront replay,
    kernels_replay.hip: k_replay_condense_narrow) and 65 / 100 pairs (the chunked kernel), grouped by arm sum as the reference walks them."""
    def by_sums(sums, e_lo, e_hi, l_lo, l_hi, cap=None):
        out = []
        for s in sums:
            out += [(e, s - e) for e in range(e_lo, e_hi + 1) if l_lo <= s - e <= l_hi]
        return out[:cap] if cap else out
    return {
        "A1": [(20, 22)],
        "A2_one_list": [(19, 23), (20, 22)],
        "A7_two_lists": by_sums([44, 41], 18, 22, 20, 24),
        "A57_default": None,
        "A63": by_sums(range(50, 40, -1), 16, 27, 18, 30, cap=63),
        "A64": by_sums(range(50, 40, -1), 16, 27, 18, 30, cap=64),
        "A65": by_sums(range(50, 40, -1), 16, 27, 18, 30, cap=65),
        "A100": by_sums(range(52, 38, -1), 16, 28, 18, 30, cap=100),
    }


@pytest.mark.parametrize("name", list(_pair_lists()))
@pytest.mark.parametrize("heuristic", [True, False])
def test_replay_condense_over_pair_list_shapes(name, heuristic):
    """Replay of the early exits (mipgen.cpp:426-497) + condense (:1670-1746) against the oracle for arm-pair lists of every shape, with the
    logistic heuristic on and off, hashed copy numbers (copy-driven takes) and an N run (guard scores) in reach."""
    pairs = _pair_lists()[name]
    genome = bytearray(H.golden_genome())
    genome[9000:9008] = b"NNNNNNNN"
    genome = bytes(genome)
    P = capi.make_params(150, 165, score_method=capi.SCORE_LOGISTIC, arm_pairs=pairs, logistic_heuristic=heuristic)
    acc = capi.Accel(P)
    rd = capi.build_region(genome, "1", 8950, 9150, P, bwa_mode="hashed", label=name)
    grids, scores, records = acc.score_regions([rd], capi.SCORE_LOGISTIC)
    acc.replay_condense()
    emitted, surv, mask = acc.download_replay()
    n_emit, omask = po.replay_region(P, rd, scores, records)
    assert emitted[0] == n_emit and np.array_equal(mask, omask), (name, heuristic)
    osurv = po.condense_region(P, rd, scores, records, omask)
    assert np.array_equal(surv["cand_index"], osurv["cand_index"]), (name, heuristic)
    assert np.array_equal(surv["record"], osurv["record"])
    assert np.array_equal(surv["score"], osurv["score"], equal_nan=True)
    if heuristic and P.n_arm_pairs > 2:
        assert n_emit < int(((capi.rec_flags(records) & capi.FLAG_VALID) != 0).sum())      # the heuristic exit did fire
    acc.close()


def _replay_cases():
    rng = np.random.default_rng(777)
    out = []
    for i in range(int(os.environ.get("MIPGEN_FUZZ_REPLAY_N", "6"))):
        n_sums = int(rng.integers(1, 8))
        sums = sorted(rng.choice(np.arange(38, 52), size=n_sums, replace=False).tolist(), reverse=True)
        pairs = []
        for s in sums:
            es = [e for e in range(16, 31) if 18 <= s - e <= 30]
            keep = rng.random(len(es)) < rng.uniform(0.3, 1.0)
            pairs += [(e, s - e) for e, k in zip(es, keep) if k]
        if not pairs:
            pairs = [(20, 22)]
        pairs = pairs[:int(rng.integers(1, 121))]
        out.append((i, pairs, int(rng.integers(3000, 16000)), int(rng.choice([30, 90, 200])), bool(rng.integers(0, 2)), int(rng.choice([1, 2, 5])),
                    int(rng.integers(120, 170))))
    return out


@pytest.mark.parametrize("case", _replay_cases(), ids=lambda c: f"r{c[0]}_A{len(c[1])}_L{c[3]}_h{int(c[4])}_inc{c[5]}")
def test_replay_condense_random(case):
    """Random arm-pair lists (1..120 pairs in 1..7 arm-sum lists), capture ranges, increments, region lengths: replay masks, emitted counts and
    condensed survivors against the oracle, for the logistic scan (heuristic on / off) and for SVR scores."""
    i, pairs, start, length, heuristic, inc, lo = case
    genome = bytearray(H.golden_genome())
    if i % 2 == 0:
        genome[start + 10:start + 15] = b"NNNNN"
    genome = bytes(genome)
    hi = lo + inc * (i % 7)
    mp = os.path.join(H.GOLDEN, "models", "svr_syn_64.model")
    for method in (capi.SCORE_LOGISTIC, capi.SCORE_SVR):
        P = capi.make_params(lo, hi, score_method=method, capture_increment=inc, arm_pairs=pairs, logistic_heuristic=heuristic)
        acc = capi.Accel(P)
        if method == capi.SCORE_SVR:
            acc.load_model_file(mp)
        rd = capi.build_region(genome, "1", start, start + length, P, bwa_mode="hashed", label=f"r{i}", lrc=np.full(44, 0.015 * (i % 5 + 1)))
        grids, scores, records = acc.score_regions([rd], method)
        acc.replay_condense()
        emitted, surv, mask = acc.download_replay()
        n_emit, omask = po.replay_region(P, rd, scores, records)
        assert emitted[0] == n_emit and np.array_equal(mask, omask), (case[0], method)
        osurv = po.condense_region(P, rd, scores, records, omask)
        assert np.array_equal(surv["cand_index"], osurv["cand_index"]), (case[0], method)
        assert np.array_equal(surv["record"], osurv["record"])
        assert np.array_equal(surv["score"], osurv["score"], equal_nan=True)
        acc.close()

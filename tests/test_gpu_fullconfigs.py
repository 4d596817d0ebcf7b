"""GPU: BASELINE.json configs[2] at its FULL size - all 1,000 regions of 5,000 bp, capture 120-250 (27 sizes x 57 arm pairs x 2 strands at
5.2 M scan starts = 1.6e10 dense candidates), the logistic scan of the mixed design - through the silent fused path
(mipgen_accel_score_condense_all: score -> replay of the early exits -> condense, window by window).  At this size the dense results
(272 GB) are produced window by window (automatic windows hold at most 2^30 candidates), so the result windows are exercised for real.  Checked through size-independent properties:

  * the survivors do not depend on how the batch is cut into result windows, nor on how the regions are sharded over handles (= ranks);
  * structural invariants of every survivor (its candidate lies in the row block of its own scan position and strand, it is valid,
    not mapping-failed, and its score is the dense score the record came with);
  * three regions drawn at random are re-scored alone, their dense grids replayed + condensed by the ORACLE
    (/root/reference/mipgen.cpp:426-497, 1670-1746): identical survivors and emitted counts.
"""
import numpy as np
import pytest

from mipgen_amd import capi, workloads
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
N_REGIONS = 1000


def _relative(surv, grids, pos0):
    """cand_index relative to the region's own grid (batch layouts differ between shardings)."""
    out = surv["cand_index"].copy()
    for g, (a, b) in zip(grids, zip(pos0[:-1], pos0[1:])):
        s = out[2 * a:2 * b]
        s[s >= 0] -= g.offset
    return out


@pytest.fixture(scope="module")
def full5k():
    genome = workloads.regions5k_genome()
    ivs = workloads.regions5k_intervals(N_REGIONS)
    P = capi.make_params(120, 250, score_method=capi.SCORE_LOGISTIC)
    regions = workloads.build_regions5k(None, genome, ivs, P, with_lrc=False)
    acc = capi.Accel(P)
    grids = acc.upload(regions)
    n_win_auto = acc.window_count()
    acc.score_condense_all(capi.SCORE_LOGISTIC)
    emitted, surv = acc.download_survivors()
    pos0 = np.concatenate([[0], np.cumsum([g.n_pos for g in grids])])
    out = dict(P=P, regions=regions, grids=grids, emitted=emitted.copy(), surv=surv.copy(), pos0=pos0, n_win_auto=n_win_auto)
    acc.close()                                            # its result windows hold most of the HBM: free them for the handles of the tests
    yield out


def test_full_config3_size_and_invariants(full5k):
    P, grids, surv, emitted, pos0 = (full5k[k] for k in ("P", "grids", "surv", "emitted", "pos0"))
    total = sum(g.count for g in grids)
    assert len(grids) == N_REGIONS and total > 1.5e10 and all(g.n_sizes == 27 for g in grids)
    assert full5k["n_win_auto"] >= 2                       # automatic windows hold at most 2^30 candidates: 1.6e10 need fifteen
    assert surv.shape[0] == 2 * pos0[-1]
    A = P.n_arm_pairs
    rel = _relative(surv, grids, pos0)
    have = rel >= 0
    assert have.mean() > 0.5
    per_pos = 27 * 2 * A
    slot = np.arange(surv.shape[0])
    pos_in_region = slot // 2 - np.repeat(pos0[:-1], 2 * np.diff(pos0))
    assert np.array_equal(rel[have] // per_pos, pos_in_region[have])             # the survivor of a scan position is one of ITS candidates
    assert np.array_equal((rel[have] // A) & 1, (slot & 1)[have])                # slot parity = strand
    flags = capi.rec_flags(surv["record"][have])
    assert np.all(flags & capi.FLAG_VALID) and not np.any(flags & capi.FLAG_MAPPING)
    # emitted <= valid dense candidates, > 0 everywhere, and the early exits did remove candidates
    assert np.all(emitted > 0) and emitted.sum() < total
    # a scan position whose '+' slot is empty has an empty '-' slot only if nothing was constructed there... the two strands are emitted together
    both = have.reshape(-1, 2)
    assert (both[:, 0] | both[:, 1]).mean() > 0.9


def test_windows_and_shards_do_not_change_the_survivors(full5k):
    P, regions, grids, pos0 = (full5k[k] for k in ("P", "regions", "grids", "pos0"))
    ref_rel = _relative(full5k["surv"], grids, pos0)
    # (a) the same batch cut into ~32 windows
    acc = capi.Accel(P)
    acc.set_window_candidates(500_000_000)
    g2 = acc.upload(regions)
    assert acc.window_count() >= 30
    acc.score_condense_all(capi.SCORE_LOGISTIC)
    e2, s2 = acc.download_survivors()
    assert np.array_equal(e2, full5k["emitted"])
    assert np.array_equal(_relative(s2, g2, pos0), ref_rel)
    assert np.array_equal(s2["record"], full5k["surv"]["record"]) and np.array_equal(s2["score"], full5k["surv"]["score"], equal_nan=True)
    acc.close()
    # (b) three contiguous shards on separate handles (what three ranks would hold), concatenated in region order
    cuts = [0, 333, 700, N_REGIONS]
    parts, emitted = [], []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        a = capi.Accel(P)
        gs = a.upload(regions[lo:hi])
        a.score_condense_all(capi.SCORE_LOGISTIC)
        e, s = a.download_survivors()
        p0 = np.concatenate([[0], np.cumsum([g.n_pos for g in gs])])
        parts.append((_relative(s, gs, p0), s["record"].copy(), s["score"].copy()))
        emitted.append(e.copy())
        a.close()
    assert np.array_equal(np.concatenate(emitted), full5k["emitted"])
    assert np.array_equal(np.concatenate([p[0] for p in parts]), ref_rel)
    assert np.array_equal(np.concatenate([p[1] for p in parts]), full5k["surv"]["record"])
    assert np.array_equal(np.concatenate([p[2] for p in parts]), full5k["surv"]["score"], equal_nan=True)


def test_sampled_regions_against_the_oracle(full5k):
    P, regions, grids, pos0, surv, emitted = (full5k[k] for k in ("P", "regions", "grids", "pos0", "surv", "emitted"))
    rng = np.random.default_rng(2024)
    acc = capi.Accel(P)
    for ri in sorted(rng.choice(N_REGIONS, 3, replace=False).tolist()):
        rd = regions[ri]
        _, scores, records = acc.score_regions([rd], capi.SCORE_LOGISTIC)
        n_emit, omask = po.replay_region(P, rd, scores, records)
        osurv = po.condense_region(P, rd, scores, records, omask)
        mine = surv[2 * pos0[ri]:2 * pos0[ri + 1]]
        assert emitted[ri] == n_emit
        assert np.array_equal(np.where(mine["cand_index"] >= 0, mine["cand_index"] - grids[ri].offset, -1), osurv["cand_index"]), ri
        assert np.array_equal(mine["record"], osurv["record"]) and np.array_equal(mine["score"], osurv["score"], equal_nan=True)
        # and the survivor's score is the dense score of its candidate
        ok = osurv["cand_index"] >= 0
        assert np.array_equal(scores[osurv["cand_index"][ok]], mine["score"][ok], equal_nan=True)
    acc.close()


# ----------------------------------------------------------------------------------------------------------------------------------
# configs[3]: the whole synthetic exome (200,000 exon-like intervals on 24 chromosomes, 300 Mb), capture 150-170, SVR n_sv = 1024
# ----------------------------------------------------------------------------------------------------------------------------------
CACHE = "/tmp/mipgen_test_cache"
TOL = 1e-5


def _decode(P, g, idx):
    A = P.n_arm_pairs
    a = idx % A
    row = idx // A
    strand = row & 1
    rest = row >> 1
    ki, pi = rest % g.n_sizes, rest // g.n_sizes
    return (0, g.first_pos + int(pi), P.max_capture_size - (g.first_size_index + int(ki)) * P.capture_increment,
            P.arm_ext[int(a)], P.arm_lig[int(a)], int(strand))


@pytest.fixture(scope="module")
def exome_full():
    chrom_len, ivs = workloads.exome_layout()
    P = capi.make_params(150, 170, score_method=capi.SCORE_SVR)
    mp = workloads.svr_model_path(CACHE, workloads.practice62()[0], 1024, rho=workloads.MODEL_RHO["exome"])   # ~12 % of the lists exit early
    acc = capi.Accel(P)
    acc.load_model_file(mp)
    regions = workloads.build_exome(acc, chrom_len, ivs, P)
    grids = acc.upload(regions)
    acc.score_condense_all(capi.SCORE_SVR)
    emitted, surv = acc.download_survivors()
    pos0 = np.concatenate([[0], np.cumsum([g.n_pos for g in grids])])
    out = dict(P=P, regions=regions, grids=grids, emitted=emitted.copy(), surv=surv.copy(), pos0=pos0, model=mp, ivs=ivs)
    acc.close()
    yield out


def test_full_exome_svr(exome_full):
    """2.5e10 dense candidates scored by the RBF-SVR in one fused pass.  Invariants of every survivor; a contiguous shard of 2,000 exons
    scored alone (another handle = another rank) gives the same survivors; three exons against the oracle (replay + condense over the
    dense grid of the exon scored alone, and the oracle's own libsvm arithmetic on sampled candidates)."""
    P, regions, grids, surv, emitted, pos0 = (exome_full[k] for k in ("P", "regions", "grids", "surv", "emitted", "pos0"))
    assert len(grids) == 200_000 and sum(g.count for g in grids) > 2.0e10
    assert len({g.n_sizes for g in grids}) >= 3                               # the static skip leaves 1..5 capture sizes on short exons
    A = P.n_arm_pairs
    rel = _relative(surv, grids, pos0)
    have = rel >= 0
    slot = np.arange(surv.shape[0])
    per_pos = np.repeat(np.array([g.n_sizes * 2 * A for g in grids]), 2 * np.diff(pos0))
    pos_in_region = slot // 2 - np.repeat(pos0[:-1], 2 * np.diff(pos0))
    assert np.array_equal(rel[have] // per_pos[have], pos_in_region[have])
    assert np.array_equal((rel[have] // A) & 1, (slot & 1)[have])
    assert np.all(capi.rec_flags(surv["record"][have]) & capi.FLAG_VALID)
    # the early exits of mipgen.cpp:430,434 are taken at exome scale (K = 1..5 capture sizes): a tenth to a half of the dense grid is never
    # constructed, and that on at least a tenth of the exons
    dense = np.array([g.count for g in grids], dtype=np.int64)
    assert 0.5 * dense.sum() < emitted.sum() < 0.9 * dense.sum(), (int(emitted.sum()), int(dense.sum()))
    assert (emitted < dense).mean() > 0.10
    # a shard alone
    lo, hi = 70_000, 72_000
    a = capi.Accel(P)
    a.load_model_file(exome_full["model"])
    gs = a.upload(regions[lo:hi])
    a.score_condense_all(capi.SCORE_SVR)
    e, s = a.download_survivors()
    p0 = np.concatenate([[0], np.cumsum([g.n_pos for g in gs])])
    assert np.array_equal(e, emitted[lo:hi])
    assert np.array_equal(_relative(s, gs, p0), rel[2 * pos0[lo]:2 * pos0[hi]])
    ref = surv[2 * pos0[lo]:2 * pos0[hi]]
    assert np.array_equal(s["record"], ref["record"])
    d = np.abs(s["score"] - ref["score"])                                     # a different SV split of the launch: sums in a different order
    assert np.nanmax(d) < 1e-9
    # the oracle on three exons
    om = po.Model(exome_full["model"])
    rng = np.random.default_rng(7)
    sizes = np.array([g.count for g in grids])
    pool = np.nonzero((sizes > 2_000) & (sizes < 400_000))[0]
    for ri in sorted(rng.choice(pool, 3, replace=False).tolist()):
        rd = regions[ri]
        g1, scores, records = a.score_regions([rd], capi.SCORE_SVR)
        n_emit, omask = po.replay_region(P, rd, scores, records)
        osurv = po.condense_region(P, rd, scores, records, omask)
        mine = surv[2 * pos0[ri]:2 * pos0[ri + 1]]
        assert emitted[ri] == n_emit, ri
        assert np.array_equal(np.where(mine["cand_index"] >= 0, mine["cand_index"] - grids[ri].offset, -1), osurv["cand_index"]), ri
        assert np.array_equal(mine["record"], osurv["record"])
        valid = np.nonzero((capi.rec_flags(records) & capi.FLAG_VALID) != 0)[0]
        lrc = np.array(rd.c.long_range_content[:])
        for idx in rng.choice(valid, size=min(40, valid.size), replace=False):
            sk, dsg = po.design(P, rd, _decode(P, g1[0], int(idx)))
            assert not sk
            so, _, _ = po.score_designed(dsg, capi.SCORE_SVR, lrc, om)
            assert abs(scores[idx] - so) <= TOL or (np.isnan(scores[idx]) and np.isnan(so)), (ri, int(idx), scores[idx], so)
    a.close()


# ----------------------------------------------------------------------------------------------------------------------------------
# configs[4]: the exome with SNP masking (1 SNP / 300 bp) and the full 120-250 capture sweep (the 4,4 smMIP tags only change the
# printed probe sequence, mipgen.cpp:790-792): 1.35e11 dense candidates, logistic scoring, ~10 result windows
# ----------------------------------------------------------------------------------------------------------------------------------
def test_full_exome_snps_full_sweep_logistic():
    chrom_len, ivs = workloads.exome_layout()
    P = capi.make_params(120, 250, score_method=capi.SCORE_LOGISTIC)
    acc = capi.Accel(P)
    regions = workloads.build_exome(None, chrom_len, ivs, P, snps=True, with_lrc=False)
    grids = acc.upload(regions)
    total = sum(g.count for g in grids)
    assert total > 1.2e11 and acc.window_count() >= 8
    acc.score_condense_all(capi.SCORE_LOGISTIC)
    emitted, surv = acc.download_survivors()
    pos0 = np.concatenate([[0], np.cumsum([g.n_pos for g in grids])])
    assert surv.shape[0] == 2 * pos0[-1] and 0 < emitted.sum() < total
    have = surv["cand_index"] >= 0
    rec = surv["record"][have]
    assert np.all(capi.rec_flags(rec) & capi.FLAG_VALID)
    assert (capi.rec_snp_count(rec) > 0).sum() > 100_000                       # condensed survivors that sit on SNPs exist ...
    assert (capi.rec_snp_count(rec) == 0).mean() > 0.5                         # ... but the fold prefers SNP-free arms (mipgen.cpp:1725)
    # shards anywhere in the batch, alone on a second handle, give the same survivors; three of their exons go through the oracle
    rng = np.random.default_rng(99)
    a = capi.Accel(P)
    for lo in (1_000, 123_456):
        hi = lo + 500
        gs = a.upload(regions[lo:hi])
        a.score_condense_all(capi.SCORE_LOGISTIC)
        e, s = a.download_survivors()
        p0 = np.concatenate([[0], np.cumsum([g.n_pos for g in gs])])
        assert np.array_equal(e, emitted[lo:hi])
        assert np.array_equal(_relative(s, gs, p0), _relative(surv[2 * pos0[lo]:2 * pos0[hi]].copy(), grids[lo:hi], pos0[lo:hi + 1] - pos0[lo]))
        assert np.array_equal(s["record"], surv["record"][2 * pos0[lo]:2 * pos0[hi]])
        assert np.array_equal(s["score"], surv["score"][2 * pos0[lo]:2 * pos0[hi]], equal_nan=True)
        sizes = np.array([g.count for g in gs])
        for k in rng.choice(np.nonzero(sizes < 3_000_000)[0], 2, replace=False).tolist():
            ri = lo + int(k)
            rd = regions[ri]
            _, scores, records = a.score_regions([rd], capi.SCORE_LOGISTIC)
            _, os_, or_ = po.score_region_dense(P, rd, capi.SCORE_LOGISTIC, None)
            assert np.array_equal(records, or_), ri
            with np.errstate(invalid="ignore"):
                assert np.all((np.abs(scores - os_) <= TOL) | (np.isnan(scores) & np.isnan(os_))), ri
            n_emit, omask = po.replay_region(P, rd, scores, records)
            osurv = po.condense_region(P, rd, scores, records, omask)
            mine = surv[2 * pos0[ri]:2 * pos0[ri + 1]]
            assert emitted[ri] == n_emit
            assert np.array_equal(np.where(mine["cand_index"] >= 0, mine["cand_index"] - grids[ri].offset, -1), osurv["cand_index"]), ri
            assert np.array_equal(mine["record"], osurv["record"])
    a.close()
    acc.close()


def test_full_exome_snps_full_sweep_svr():
    """configs[4] at its full size WITH the SVR (1.35e11 dense candidates, 1,024 support vectors: ~70 s of k_svr_dense over ten-odd result
    windows): the same invariants as the logistic sweep, early exits taken (rho placed for this workload), a shard alone on a second handle
    gives the same survivors, and three exons go through the oracle (replay + condense of the exon's dense grid, and the oracle's own libsvm
    arithmetic on sampled candidates)."""
    chrom_len, ivs = workloads.exome_layout()
    P = capi.make_params(120, 250, score_method=capi.SCORE_SVR)
    mp = workloads.svr_model_path(CACHE, workloads.practice62()[0], 1024, rho=workloads.MODEL_RHO["exome_snp"])
    acc = capi.Accel(P)
    acc.load_model_file(mp)
    regions = workloads.build_exome(acc, chrom_len, ivs, P, snps=True)
    grids = acc.upload(regions)
    total = sum(g.count for g in grids)
    assert total > 1.2e11 and acc.window_count() >= 8
    acc.score_condense_all(capi.SCORE_SVR)
    emitted, surv = acc.download_survivors()
    emitted, surv = emitted.copy(), surv.copy()
    acc.close()                                                               # its result windows hold most of the HBM
    pos0 = np.concatenate([[0], np.cumsum([g.n_pos for g in grids])])
    assert surv.shape[0] == 2 * pos0[-1]
    dense = np.array([g.count for g in grids], dtype=np.int64)
    assert 0.2 * total < emitted.sum() < 0.8 * total and (emitted < dense).mean() > 0.10      # the early exits are taken
    have = surv["cand_index"] >= 0
    rec = surv["record"][have]
    assert np.all(capi.rec_flags(rec) & capi.FLAG_VALID)
    assert (capi.rec_snp_count(rec) > 0).sum() > 100_000 and (capi.rec_snp_count(rec) == 0).mean() > 0.5
    A = P.n_arm_pairs
    rel = _relative(surv, grids, pos0)
    slot = np.arange(surv.shape[0])
    per_pos = np.repeat(np.array([g.n_sizes * 2 * A for g in grids]), 2 * np.diff(pos0))
    pos_in_region = slot // 2 - np.repeat(pos0[:-1], 2 * np.diff(pos0))
    assert np.array_equal(rel[have] // per_pos[have], pos_in_region[have])
    assert np.array_equal((rel[have] // A) & 1, (slot & 1)[have])
    # a shard alone (another handle = another rank), then three of its exons against the oracle
    om = po.Model(mp)
    rng = np.random.default_rng(4)
    a = capi.Accel(P)
    a.load_model_file(mp)
    lo, hi = 150_000, 150_400
    gs = a.upload(regions[lo:hi])
    a.score_condense_all(capi.SCORE_SVR)
    e, s = a.download_survivors()
    p0 = np.concatenate([[0], np.cumsum([g.n_pos for g in gs])])
    assert np.array_equal(e, emitted[lo:hi])
    assert np.array_equal(_relative(s, gs, p0), rel[2 * pos0[lo]:2 * pos0[hi]])
    ref = surv[2 * pos0[lo]:2 * pos0[hi]]
    assert np.array_equal(s["record"], ref["record"])
    assert np.nanmax(np.abs(s["score"] - ref["score"])) < 1e-9               # another SV split of the launch: sums in another order
    sizes = np.array([g.count for g in gs])
    for k in rng.choice(np.nonzero((sizes > 50_000) & (sizes < 1_500_000))[0], 3, replace=False).tolist():
        ri = lo + int(k)
        rd = regions[ri]
        g1, scores, records = a.score_regions([rd], capi.SCORE_SVR)
        n_emit, omask = po.replay_region(P, rd, scores, records)
        osurv = po.condense_region(P, rd, scores, records, omask)
        mine = surv[2 * pos0[ri]:2 * pos0[ri + 1]]
        assert emitted[ri] == n_emit, ri
        assert np.array_equal(np.where(mine["cand_index"] >= 0, mine["cand_index"] - grids[ri].offset, -1), osurv["cand_index"]), ri
        assert np.array_equal(mine["record"], osurv["record"])
        valid = np.nonzero((capi.rec_flags(records) & capi.FLAG_VALID) != 0)[0]
        lrc = np.array(rd.c.long_range_content[:])
        for idx in rng.choice(valid, size=min(40, valid.size), replace=False):
            sk, dsg = po.design(P, rd, _decode(P, g1[0], int(idx)))
            assert not sk
            so, _, oints = po.score_designed(dsg, capi.SCORE_SVR, lrc, om)
            assert abs(scores[idx] - so) <= TOL or (np.isnan(scores[idx]) and np.isnan(so)), (ri, int(idx), scores[idx], so)
    a.close()


def test_full_config3_mixed_command_line_three_routes(tmp_path):
    """BASELINE configs[2] as its COMMAND LINE at full size: `mipgen -score_method mixed -silent_mode on` over all 1,000 regions of 5,000 bp, capture
    120-250 - 1.6e10 candidates scanned with the logistic score, ~1e7 condensed survivors re-scored with the 1,024-SV SVR on the device, the pick stage
    on the host (mipgen.cpp:503-520, 1523-1527, 1873-1877).  Three independent routes of the product must write the same picked file: the in-process
    front end with per-GPU PCIe downloads, the same with `-gpu_gather rccl` (one grouped RCCL send / receive per result window into GPU 0), and
    mipgen_amd/mp_design.py with two ranks (one process per rank, torch.distributed gather; gloo on this one-GPU box)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    work = str(tmp_path / "c3")
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "cli_exome.py"), "1000", work, "regions5k", "mixed"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1200)
    assert p.returncode == 0, (p.stdout.decode()[-2000:], p.stderr.decode()[-2000:])
    picked = open(os.path.join(work, "out.picked_mips.txt"), "rb").read()
    lines = picked.split(b"\n")[1:-1]
    assert len(lines) > 40_000                                                  # ~53,000 MIPs tile the 5 Mb of targets
    sc = np.array([float(l.split(b"\t")[1]) for l in lines[:5000]])
    assert np.isfinite(sc).all() and lines[0].split(b"\t")[2] == b"1"
    argv = [os.path.join(work, "mipgen"), "-regions_to_scan", os.path.join(work, "exome.bed"), "-project_name", "out2", "-min_capture_size", "120", "-max_capture_size", "250",
            "-bwa_genome_index", os.path.join(work, "genome", "index.fa"), "-genome_dir", os.path.join(work, "genome"), "-score_method", "mixed", "-silent_mode", "on",
            "-gpu_copy_counter", "on"]
    q = subprocess.run(argv + ["-gpus", "1", "-gpu_gather", "rccl", "-gpu_timing", "on"], cwd=work, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1200)
    assert q.returncode == 0 and b"[mipgen timing] rccl gather:" in q.stderr, q.stderr.decode()[-2000:]
    assert open(os.path.join(work, "out2.picked_mips.txt"), "rb").read() == picked
    argv[argv.index("out2")] = "out3"
    env = dict(os.environ, PYTHONPATH=root)
    r = subprocess.run([sys.executable, "-m", "mipgen_amd.mp_design", "--gpus", "2", "--backend", "gloo", "--share-gpus", "--mipgen-path", argv[0], "--"] + argv[1:],
                       cwd=work, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1800)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    assert open(os.path.join(work, "out3.picked_mips.txt"), "rb").read() == picked

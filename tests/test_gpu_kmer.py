"""GPU: SURVEY.md section 8f-3 - arm-oligo copy numbers by exact k-mer counting on the device (opt-in replacement of the reference's
FASTQ -> bwa aln / samse -> X0:i round trip, /root/reference/mipgen.cpp:558-596, 825-835).  Parity against BWA itself is unpinned (no bwa in
this image, and X0 counts mismatch-tolerant best hits); the device path is held bit-exact to the dictionary counter in oracle/pyoracle.py."""
import os
import subprocess

import numpy as np
import pytest

from mipgen_amd import capi, synth
from oracle import pyoracle as po
from tests import helpers as H

pytestmark = pytest.mark.gpu
_RC = bytes.maketrans(b"ACGT", b"TGCA")


def _genome_with_repeats(seed=3):
    rng = np.random.default_rng(seed)
    a = bytearray(synth.random_genome(60_000, 31, n_run_frac=0.002, n_run_len=7))
    b = bytearray(synth.random_genome(40_000, 32))
    unit = bytes(a[5000:5300])
    for pos in (12_000, 30_500, 51_000):                      # direct repeats of a 300-base unit
        a[pos:pos + 300] = unit
    b[7000:7300] = unit
    b[20_000:20_300] = unit.translate(_RC)[::-1]              # and an inverted copy on the other chromosome
    a[41_000:41_060] = b"ACGT" * 15                           # low-complexity stretch: palindromic k-mers
    for i in rng.integers(0, 60_000, 40):
        a[int(i)] = ord("acgtn"[int(i) % 5])                  # soft-masked (lower case) and n bytes
    return bytes(a), bytes(b)


def test_copy_counter_vs_dictionary_counter():
    g1, g2 = _genome_with_repeats()
    P = capi.make_params(152, 162)
    acc = capi.Accel(P)
    lengths = sorted({e for e, _ in capi.arm_pairs_of(P)} | {l for _, l in capi.arm_pairs_of(P)})
    assert lengths[0] == 16 and lengths[-1] == 29
    regions = [g1[4800:5500].upper(), g1[11_900:12_400].upper(), g1[40_950:41_120].upper(), g2[19_900:20_400].upper(), g2[100:160].upper(), b"ACGTN" * 8]
    got = acc.count_oligo_copies([g1, g2], regions, lengths)
    for seq, tab in zip(regions, got):
        exp = po.count_oligo_copies([g1, g2], seq, lengths)
        for k in lengths:
            assert np.array_equal(tab[k], exp[k]), (k, np.nonzero(tab[k] != exp[k])[0][:5], tab[k][:8], exp[k][:8])
    assert max(int(t[16].max()) for t in got) >= 6            # the planted unit: 6 copies (incl. the inverted one)
    assert acc.last_kernel_ms(4) > 0
    with pytest.raises(capi.AccelError):
        acc.count_oligo_copies([g1], regions, [16, 40])       # > 31: no exact 2-bit key
    acc.close()


def test_copy_counter_large_design_takes_the_unfolded_filter():
    """More than 2^16 region positions: the LDS fold of the Bloom filter is skipped and every genome position tests the full bitmap
    (kernels_kmer.hip: use_fold); same counts as the dictionary counter."""
    g1, g2 = _genome_with_repeats(seed=5)
    P = capi.make_params(152, 162)
    acc = capi.Accel(P)
    lengths = sorted({e for e, _ in capi.arm_pairs_of(P)} | {l for _, l in capi.arm_pairs_of(P)})
    regions = [g1[i:i + 2000].upper() for i in range(0, 58_000, 2000)] + [g2[i:i + 2000].upper() for i in range(0, 38_000, 2000)]
    assert sum(len(r) + 1 for r in regions) * 32 > 1 << 21
    got = acc.count_oligo_copies([g1, g2], regions, lengths)
    for ri in (2, 6, 15, 20, 39):                             # incl. the regions holding the planted repeats
        exp = po.count_oligo_copies([g1, g2], regions[ri], lengths)
        for k in lengths:
            assert np.array_equal(got[ri][k], exp[k]), (ri, k, np.nonzero(got[ri][k] != exp[k])[0][:5])
    acc.close()


def test_resident_copy_tables_equal_host_tables():
    """mipgen_accel_count_oligo_copies_resident + upload with MIPGEN_COPY_RESIDENT: the counts stay in HBM in the layout the kernels read.
    Same records, scores and survivors as the tables that went through the host; the entries of 65535 copies and more come back as a list."""
    g1, g2 = _genome_with_repeats(seed=7)
    rng = np.random.default_rng(11)
    g3 = bytes(synth.random_genome(3000, 41)) + b"A" * 70_000 + bytes(synth.random_genome(3000, 42))          # 16..29-mers with > 65535 copies
    chroms = [g1, g2, g3]
    P = capi.make_params(152, 162)
    lengths = sorted({e for e, _ in capi.arm_pairs_of(P)} | {l for _, l in capi.arm_pairs_of(P)})
    spans = [(g1, 5200, 5500), (g1, 12_050, 12_300), (g2, 7100, 7250), (g3, 2900, 3100), (g3, 72_950, 73_200), (g2, 30_000, 30_040)]
    base = [capi.build_region(g, "1", a, b, P, flank=int(rng.integers(0, 4))) for g, a, b in spans]
    seqs = [r.seq for r in base]

    def regions(copy_of):
        return [capi.RegionData(r.c.start_flanked, r.c.stop_flanked, r.c.seq_start, r.seq, copy=copy_of(i)) for i, r in enumerate(base)]

    acc = capi.Accel(P)
    tabs = acc.count_oligo_copies(chroms, seqs, lengths)
    _, s_host, r_host = acc.score_regions(regions(lambda i: tabs[i]), capi.SCORE_LOGISTIC)
    acc.replay_condense()
    _, surv_host = acc.download_survivors()
    acc.collapse()
    col_host = acc.download_collapsed()
    acc.close()

    acc = capi.Accel(P)
    big = acc.count_oligo_copies_resident(chroms, seqs)
    _, s_res, r_res = acc.score_regions(regions(lambda i: capi.COPY_RESIDENT), capi.SCORE_LOGISTIC)
    acc.replay_condense()
    _, surv_res = acc.download_survivors()
    acc.collapse()
    col_res = acc.download_collapsed()
    assert np.array_equal(r_host, r_res) and np.array_equal(s_host, s_res, equal_nan=True)
    assert surv_host.tobytes() == surv_res.tobytes() and np.array_equal(col_host, col_res)
    exp_big = sorted((ri, k, int(i), int(t[k][i])) for ri, t in enumerate(tabs) for k in lengths for i in np.nonzero(t[k] >= 65535)[0])
    assert big == exp_big and len(big) > 100
    assert int((capi.rec_ext_copy(r_res) == 65535).sum()) > 0                    # saturated record fields: the host reads the list
    # the resident tables belong to one batch: another order, another handle or a partly resident batch is refused
    with pytest.raises(capi.AccelError):
        acc.upload(regions(lambda i: capi.COPY_RESIDENT)[::-1])
    with pytest.raises(capi.AccelError):
        acc.upload(regions(lambda i: capi.COPY_RESIDENT if i else tabs[0]))
    acc.upload(regions(lambda i: capi.COPY_RESIDENT))                            # still there after the refused uploads
    acc.upload(regions(lambda i: tabs[i]))                                       # host tables overwrite them ...
    with pytest.raises(capi.AccelError):
        acc.upload(regions(lambda i: capi.COPY_RESIDENT))                        # ... so the resident batch is gone
    acc.close()
    fresh = capi.Accel(P)
    with pytest.raises(capi.AccelError):
        fresh.upload(regions(lambda i: capi.COPY_RESIDENT))
    fresh.close()


def test_cli_with_gpu_copy_counter(tmp_path):
    """`mipgen ... -gpu_copy_counter on`: no bwa is run (the -bwa path does not even exist), the copy columns of all_mips are the exact
    occurrence counts of the printed arm sequences in the genome, and the design still tiles."""
    meta = H.load_design("logistic_default_arms")
    work = str(tmp_path)
    argv = H.prepare_cli_workdir(meta, work)
    argv[argv.index("-bwa") + 1] = "/nonexistent/bwa"
    p = subprocess.run(argv + ["-gpu_copy_counter", "on"], cwd=work, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    assert not os.path.exists(os.path.join(work, "out.oligo_copy_count.fq"))
    genome = H.golden_genome()
    lines = open(os.path.join(work, "out.all_mips.txt"), "rb").read().split(b"\n")[1:-1]
    assert len(lines) > 100_000
    rng = np.random.default_rng(1)
    cache = {}

    def count(s):
        if s not in cache:
            r = s.translate(_RC)[::-1]
            n, k = 0, len(s)
            for i in range(len(genome) - k + 1):
                w = genome[i:i + k]
                n += (w == s) or (w == r and r != s)
            cache[s] = n
        return cache[s]
    for i in rng.choice(len(lines), 60, replace=False):
        f = lines[int(i)].split(b"\t")
        ext_seq, lig_seq, ext_copy, lig_copy = f[6], f[10], int(f[5]), int(f[9])
        for seq, c in ((ext_seq, ext_copy), (lig_seq, lig_copy)):
            if b"N" in seq:
                assert c == 100
            else:
                assert c == count(seq), (seq, c, count(seq))
    assert open(os.path.join(work, "out.picked_mips.txt"), "rb").read().count(b"\n") >= 5


def test_front_end_counts_where_the_regions_are_scored(tmp_path):
    """-gpu_copy_counter on: every device worker of tile_regions counts its own shard and keeps the tables in its HBM - two workers write
    the same files as one; a caller that asks for the regions (mipgen_design_region) gets host tables, equal to the dictionary counter."""
    import ctypes as C
    from mipgen_amd import hostapi
    from tests import host_select_common as HS
    meta = H.load_design("logistic_default_arms")
    outs = {}
    for gpus in ("1", "2"):
        work = str(tmp_path / ("w" + gpus))
        argv = H.prepare_cli_workdir(meta, work)
        argv[argv.index("-bwa") + 1] = "/nonexistent/bwa"
        p = subprocess.run(argv + ["-gpu_copy_counter", "on", "-gpus", gpus, "-gpu_timing", "on"], cwd=work, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           timeout=600)
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        assert b"arm copy numbers (resident)" in p.stderr
        outs[gpus] = {k: open(os.path.join(work, "out." + k + ".txt"), "rb").read() for k in ("all_mips", "collapsed_mips", "picked_mips", "snp_mips")}
    assert outs["1"] == outs["2"] and outs["1"]["all_mips"].count(b"\n") > 100_000
    work = str(tmp_path / "w3")
    argv = H.prepare_cli_workdir(meta, work)
    argv[argv.index("-bwa") + 1] = "/nonexistent/bwa"
    d = HS.open_design(argv + ["-gpu_copy_counter", "on"], work)
    genome = H.golden_genome()
    P = d.params()
    lengths = sorted({e for e, _ in capi.arm_pairs_of(P)} | {l for _, l in capi.arm_pairs_of(P)})
    for i in (d.region_count() - 1, 0):
        r = d.region(i)
        assert bool(r.copy) and C.cast(r.copy, C.c_void_p).value != 1
        seq = r.seq[:r.seq_len]
        exp = po.count_oligo_copies([genome], seq, lengths)
        for k in lengths:
            got = np.ctypeslib.as_array(r.copy[k], shape=(r.seq_len,))
            assert np.array_equal(got, exp[k]), (i, k)
    d.close()

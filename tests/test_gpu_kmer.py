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


@pytest.mark.parametrize("lengths", [[8, 11, 15], [12, 16, 17], [17, 22, 31], [31], [16]])
def test_copy_counter_key_widths_and_chunk_borders(lengths):
    """The genome pass rolls 32-bit keys when the shortest length is <= 16 and 64-bit keys above, places the first kmin - 1 bases of a thread's
    16 window starts directly from the packed chunk, and stages 8,192 starts per pass: regions are planted across chunk and thread borders, next to
    N runs and at the very end of a chromosome."""
    g1 = bytearray(synth.random_genome(3 * 8192 + 1000, 41, n_run_frac=0.001, n_run_len=5))
    g2 = bytearray(synth.random_genome(9000, 42))
    unit = bytes(g1[8192 - 40:8192 + 60])                     # straddles the first chunk border
    g1[16_384 - 7:16_384 + 93] = unit                         # a second copy across the next border, at another thread phase
    g2[-100:] = unit                                          # a third one ends with the chromosome
    g2[4000:4003] = b"NNN"
    g1, g2 = bytes(g1), bytes(g2)
    regions = [g1[8192 - 60:8192 + 80].upper(), g2[3950:4060].upper(), g2[-130:].upper(), g1[16_000:16_030].upper()]
    acc = capi.Accel(capi.make_params(152, 162))
    got = acc.count_oligo_copies([g1, g2], regions, lengths)
    for seq, tab in zip(regions, got):
        exp = po.count_oligo_copies([g1, g2], seq, lengths)
        for k in lengths:
            assert np.array_equal(tab[k], exp[k]), (k, np.nonzero(tab[k] != exp[k])[0][:5], tab[k][:8], exp[k][:8])
    assert max(int(t[lengths[0]].max()) for t in got) >= 3
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


# ---- capture-window uniqueness (mapping_failed) without bwa: mipgen_accel_window_uniqueness vs the brute-force Hamming counter ------------------

def _genome_with_window_repeats():
    """One 80 kb chromosome + a 30 kb one with planted copies of pieces of the test region g[20_000:21_000]:
    an exact forward copy, an inverted copy with ONE substitution, a copy with two substitutions 90 bases apart (windows that hold one of
    them are X1 hits, windows that hold both are not hits at all), a copy whose substitution sits in the window's FIRST seed (found through the
    second seed only), eleven more exact copies of one piece (X0 = 12: the reference's substring test "X0:i:1" accepts it), an N, lower case,
    and a reverse-palindromic 30-mer."""
    a = bytearray(synth.random_genome(80_000, 41))
    b = bytearray(synth.random_genome(30_000, 42))
    pal = bytes(a[20_300:20_315])
    a[20_315:20_330] = pal.translate(_RC)[::-1]                       # a[20300:20330] reads the same on both strands

    def sub(seg, *at):
        seg = bytearray(seg)
        for i in at:
            seg[i] = ord("A") if seg[i] != ord("A") else ord("C")
        return bytes(seg)
    a[50_000:50_260] = a[20_040:20_300]                               # exact forward copy
    b[5_000:5_250] = sub(a[20_350:20_600], 125).translate(_RC)[::-1]  # inverted, one substitution
    a[60_000:60_300] = sub(a[20_600:20_900], 100, 190)                # two substitutions 90 apart
    b[12_000:12_200] = sub(a[20_100:20_300], 10)                      # substitution inside the first 30-mer of the windows starting at +0..+10
    for k in range(11):
        b[15_000 + 300 * k:15_000 + 300 * k + 170] = a[20_820:20_990]  # eleven more exact copies
    a[20_700] = ord("N")
    for i in range(20_500, 20_520):
        a[i] = ord(chr(a[i]).lower())
    return bytes(a), bytes(b)


def test_window_uniqueness_vs_brute_force():
    g1, g2 = _genome_with_window_repeats()
    P = capi.make_params(120, 180)
    acc = capi.Accel(P)
    sizes = [180, 150, 120]
    regions = [g1[19_800:21_200].upper(), g2[4_900:5_400].upper(), g1[100:400].upper(), b"ACGT" * 40]
    got = acc.window_uniqueness([g1, g2], regions, sizes, seed_len=30)
    n_flagged = 0
    for seq, tab in zip(regions, got):
        exp = po.window_unmappable([g1, g2], seq, sizes)
        for c, size in enumerate(sizes):
            f, x0, x1 = exp[size]
            bad = np.nonzero(tab[c] != f)[0]
            assert bad.size == 0, (size, bad[:8], tab[c][bad[:8]], f[bad[:8]], x0[bad[:8]], x1[bad[:8]])
            n_flagged += int(f.sum())
    exp0 = po.window_unmappable([g1, g2], regions[0], [120])[120]
    assert (exp0[1] == 2).any() and (exp0[2] >= 1).any() and (exp0[1] == 12).any()      # the planted cases really occur: X0 = 2, X1 > 0, X0 = 12
    assert int(exp0[0][(exp0[1] == 12) & (exp0[2] == 0)].sum()) == 0                      # "X0:i:12" contains "X0:i:1": accepted (mipgen.cpp:852)
    assert n_flagged > 500
    # a shorter seed finds the same windows (the pigeonhole holds for any seed <= size / 2)
    got20 = acc.window_uniqueness([g1, g2], regions[:2], sizes, seed_len=20)
    for t30, t20 in zip(got, got20):
        assert np.array_equal(t30, t20)
    # seeds of <= 16 bases are 32-bit keys: the Bloom bit is set and tested with the 32-bit hash (the verify pass once tested the 64-bit one:
    # nearly every locus of a repeated seed was dropped, X0 = 0, every window flagged)
    for k in (16, 12):
        gotk = acc.window_uniqueness([g1, g2], regions[:2], sizes, seed_len=k)
        for t30, tk in zip(got, gotk):
            assert np.array_equal(t30, tk), k
    with pytest.raises(capi.AccelError):
        acc.window_uniqueness([g1], regions, [50], seed_len=30)                           # a window must hold two disjoint seeds
    # the bounded form (what the front end uses): the restriction to the window starts the reference looks up (mipgen.cpp:808-813) happens on
    # the device, and only the regions with a flagged start hand out an image
    bounds = [(200 + 180, len(regions[0]) - 400, 1, len(regions[0])),                   # a region string that starts at chromosome coordinate 1
              (1_000_000, 1_000_000 + 60, 1_000_000 - 180, 1_000_000 - 180 + len(regions[1]) - 1),
              (50, 60, 1, len(regions[2])), (10, 20, 1, len(regions[3]))]
    any_, imgs = acc.window_uniqueness_bounded([g1, g2], regions, bounds, sizes, seed_len=30)
    for r, (seq, full) in enumerate(zip(regions, got)):
        sf, ef, s0, s1 = bounds[r]
        exp = full.copy()
        pos = s0 + np.arange(len(seq))
        for c, size in enumerate(sizes):
            keep = (pos >= sf - size) & (pos < ef) & (pos > 0) & (pos + size - 1 <= s1)
            exp[c][~keep] = 0
        assert bool(any_[r]) == bool(exp.any()), r
        if exp.any():
            assert np.array_equal(imgs[r], exp), r
        else:
            assert imgs[r] is None
    assert any_.sum() >= 2
    acc.close()


def test_cli_mapping_flag_from_gpu_counter(tmp_path):
    """`-gpu_copy_counter on` sets mapping_failed (first character of the failure_flags column, mipgen.cpp:615-625,791) for every candidate
    whose footprint starts at a window start that is not unique within one substitution - checked record by record against the brute-force
    counter; with -check_copy_number off the flag stays 0."""
    g1, g2 = _genome_with_window_repeats()
    work = str(tmp_path)
    os.makedirs(os.path.join(work, "genome"))
    synth.write_fasta(os.path.join(work, "genome", "chr1.fa"), "chr1", g1.upper())
    synth.write_fasta(os.path.join(work, "genome", "chr2.fa"), "chr2", g2.upper())       # carries no region: its copies must still be seen
    with open(os.path.join(work, "regions.bed"), "w") as fh:
        fh.write("chr1\t20100\t20260\tamp1\n")
    exe = os.path.join(work, "mipgen")
    os.symlink(H.CLI_BIN, exe)
    argv = [exe, "-regions_to_scan", os.path.join(work, "regions.bed"), "-project_name", "out", "-min_capture_size", "120", "-max_capture_size", "130",
            "-bwa_genome_index", os.path.join(work, "genome", "index.fa"), "-bwa", "/nonexistent/bwa", "-genome_dir", os.path.join(work, "genome"),
            "-arm_length_sums", "44,45", "-gpu_copy_counter", "on"]
    p = subprocess.run(argv, cwd=work, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    assert b"chr*.fa file(s)" in p.stderr                                                 # it says what it counted against
    lines = open(os.path.join(work, "out.all_mips.txt"), "rb").read().split(b"\n")[1:-1]
    assert len(lines) > 10_000
    # the region string of the design: [start_fl - maxC, stop_fl + maxC + 14] (mipgen.cpp:1200-1220), 1-based
    seq_start = 20_101 - 130
    seq = g1.upper()[seq_start - 1:20_260 + 130 + 15]
    exp = po.window_unmappable([g1.upper(), g2.upper()], seq, [130, 125, 120])
    n_failed = 0
    for ln in lines:
        f = ln.split(b"\t")
        ext_start, ext_stop, lig_start, lig_stop, strand, flags = int(f[3]), int(f[4]), int(f[7]), int(f[8]), f[17], f[18]
        first, last = min(ext_start, lig_start), max(ext_stop, lig_stop)
        C = last - first + 1
        mip_start = ext_start if strand == b"+" else lig_start                            # get_mip_start(): Plus/MinusSVMipv4
        assert mip_start == first
        want = int(exp[C][0][mip_start - seq_start])
        assert flags[:1] == (b"1" if want else b"0"), (ln[:200], want)
        n_failed += want
    assert n_failed > 200 and n_failed < len(lines)
    # the copies on chr2 (no region there) are counted: an arm inside the eleven-fold repeat would show them - here: the oligo copy of the
    # exact forward copy at chr1:50,000 and the one-substitution copy on chr2 are both needed to flag these windows
    p = subprocess.run(argv + ["-check_copy_number", "off", "-project_name", "off"], cwd=work, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    off = open(os.path.join(work, "off.all_mips.txt"), "rb").read().split(b"\n")[1:-1]
    assert off and all(ln.split(b"\t")[18][:1] == b"0" for ln in off)

"""GPU: SURVEY.md section 8f-3 under a repeat family.  The opt-in capture-window uniqueness test (kernels_window.hip: seed-and-extend from the k-mer
table; replaces the bwa round trip of /root/reference/mipgen.cpp:806-823, 841-868) does work proportional to (region positions of a seed) x (genome
loci of that seed): an interspersed family (Alu-like: 300 bp, thousands of copies, 10-15 % diverged, both orientations), a low-divergence tandem
satellite and a microsatellite are what stresses it - the seed lists, the per-window atomics, the extension loops.  Bit-exact against the brute-force
definition (oracle/mipgen_oracle.c: mo_window_unmappable) on a 1.5 Mb genome, and bounded time + consistent flags at 48 Mb with 10 % of the genome
in the family."""
import json
import os
import time

import numpy as np
import pytest

from mipgen_amd import capi, synth
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
_RC = bytes.maketrans(b"ACGT", b"TGCA")


def _mutate(rng, seq: bytes, frac: float) -> bytes:
    a = np.frombuffer(seq, dtype=np.uint8).copy()
    hit = np.nonzero(rng.random(a.shape[0]) < frac)[0]
    a[hit] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, hit.shape[0])]
    return a.tobytes()


def repeat_genome(n: int, seed: int, alu_frac: float, sat_copies: int, n_micro: int):
    """iid genome of n bases with (a) an Alu-like family: one 300-bp consensus, copies diverged by 10-15 %, either orientation, covering alu_frac of
    the genome; (b) a tandem satellite: a 171-bp monomer, sat_copies times head to tail, 2 % diverged per copy; (c) n_micro (CA)n runs of 400 bp.
    Returns the genome and the start positions of the planted elements."""
    rng = np.random.default_rng(seed)
    g = bytearray(synth.random_genome(n, seed))
    alu = synth.random_genome(300, seed + 1)
    n_alu = int(n * alu_frac / 300)
    alu_pos = np.sort(rng.choice(np.arange(2000, n - 2000, 400), size=n_alu, replace=False)) if n_alu else np.zeros(0, dtype=np.int64)
    for p in alu_pos:
        c = _mutate(rng, alu, float(rng.uniform(0.10, 0.15)))
        if rng.random() < 0.5:
            c = c.translate(_RC)[::-1]
        g[int(p):int(p) + 300] = c
    mono = synth.random_genome(171, seed + 2)
    sat0 = n // 3
    for k in range(sat_copies):
        g[sat0 + 171 * k:sat0 + 171 * (k + 1)] = _mutate(rng, mono, 0.02)
    micro_pos = [int(x) for x in rng.choice(np.arange(2 * n // 3, n - 5000, 2000), size=n_micro, replace=False)] if n_micro else []
    for p in micro_pos:
        g[p:p + 400] = b"CA" * 200
    return bytes(g), [int(p) for p in alu_pos], sat0, micro_pos


def test_window_uniqueness_in_a_repeat_family_vs_brute_force():
    g, alu_pos, sat0, micro = repeat_genome(1_500_000, 77, alu_frac=0.10, sat_copies=150, n_micro=6)
    sizes = [160, 120]
    mid = alu_pos[len(alu_pos) // 2]
    regions = [g[mid - 60:mid + 330].upper(),                  # an Alu copy with its unique flanks
               g[sat0 + 171 * 40 - 30:sat0 + 171 * 42 + 60].upper(),   # inside the satellite
               g[micro[0] - 120:micro[0] + 260].upper(),       # unique sequence running into a (CA)n microsatellite
               g[700_000:700_300].upper(),                      # (mostly) unique
               g[alu_pos[3] + 150:alu_pos[3] + 500].upper()]   # the right half of another copy + flank
    acc = capi.Accel(capi.make_params(120, 160))
    t0 = time.perf_counter()
    got = acc.window_uniqueness([g], regions, sizes, seed_len=30)
    dt = time.perf_counter() - t0
    flagged = 0
    for seq, tab in zip(regions, got):
        exp = po.window_unmappable([g], seq, sizes)
        for c, size in enumerate(sizes):
            f, x0, x1 = exp[size]
            bad = np.nonzero(tab[c] != f)[0]
            assert bad.size == 0, (size, bad[:8], tab[c][bad[:8]], f[bad[:8]], x0[bad[:8]], x1[bad[:8]])
            flagged += int(f.sum())
    sat = po.window_unmappable([g], regions[1], [120])[120]
    mic = po.window_unmappable([g], regions[2], [120])[120]
    assert flagged > 100 and int(mic[1].max()) > 50 and dt < 30.0, (flagged, int(sat[1].max()), int(mic[1].max()), dt)   # the microsatellite windows have hundreds of loci
    acc.close()


def test_window_uniqueness_at_scale_with_ten_percent_of_the_genome_in_a_family():
    """48 Mb, 16,000 Alu-like copies, a 1,000-copy satellite, 40 microsatellites; 2,000 regions of 250 bp of which a quarter start inside a family
    copy and a few sit in the satellite / a microsatellite / an exact 600-bp duplication.  Bounded time (the work is proportional to the repeated
    seeds of the DESIGN, not to the family size), flags consistent with what was planted, and the time against the family's share of the genome."""
    sizes = [170, 150]
    out = {}
    for frac in (0.0, 0.10):
        g, alu_pos, sat0, micro = repeat_genome(48_000_000, 91, alu_frac=frac, sat_copies=1000, n_micro=40)
        gb = bytearray(g)
        gb[5_000_000:5_000_600] = gb[9_000_000:9_000_600]        # an exact 600-bp duplication
        g = bytes(gb)
        rng = np.random.default_rng(5)
        starts = [int(x) for x in rng.integers(100_000, 47_000_000, 1500)]
        if alu_pos:
            starts += [int(alu_pos[int(i)]) - 100 for i in rng.integers(0, len(alu_pos), 490)]
        starts += [sat0 + 171 * 300, sat0 + 171 * 700 + 40, micro[0] - 100, micro[5] + 50, micro[9] + 120, 5_000_100, 9_000_200]
        regs = [g[s:s + 250].upper() for s in starts]
        bounds = [(s + 1 + 170, s + 250 - 30, s + 1, s + 250) for s in starts]
        acc = capi.Accel(capi.make_params(150, 170))
        t0 = time.perf_counter()
        any_, imgs = acc.window_uniqueness_bounded([g], regs, bounds, sizes, seed_len=30)
        dt = time.perf_counter() - t0
        acc.close()
        n = len(starts)
        # the planted cases: the duplication (both copies), the microsatellites and the low-divergence satellite are flagged; most random regions are not
        assert any_[n - 1] and any_[n - 2] and any_[n - 4] and any_[n - 3], any_[-7:]
        assert imgs[n - 2] is not None and imgs[n - 2].any()
        assert float(np.mean(any_[:1500])) < 0.25, float(np.mean(any_[:1500]))
        out[f"family_share_{frac:.2f}"] = {"seconds": dt, "regions": n, "regions_with_a_flag": int(any_.sum()), "family_copies": len(alu_pos)}
        assert dt < 60.0, out
    d = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "gpurun_out")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "r05_window_uniqueness_repeats.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    # a tenth of the genome in a diverged family costs little: diverged copies share few exact 30-mers, the work follows the design's repeated seeds
    assert out["family_share_0.10"]["seconds"] < 3.0 * out["family_share_0.00"]["seconds"] + 5.0, out

"""Synthetic stand-ins for the BASELINE.json workloads, built with the product's own
host-side helpers.  Used by bench.py and the full-size tests.  Nothing here touches the oracle.

  practice62     configs[0]/[1]: 62 exon-like regions on a 400 kb chromosome "7" (stand-in for practice_genes.bed)
  regions5k      configs[2]: 1,000 x 5,000 bp regions at stride 12 kb on a 12 Mb chromosome "1", 0.1 % of bases N in runs
                 of 50, per-oligo copy table 98 % 1 / 1.5 % U{2..20} / 0.5 % 101
  exome200k      configs[3]: 200,000 exon-like intervals (log-normal lengths, median 130, sigma 0.8, clipped [20, 10000]) on 24
                 chromosomes totalling 300 Mb at 41 % GC
  exome200k_snp  configs[4]: the same + 1 SNP per 300 bp (5 % multi-base references) for the SNP classes; -tag_sizes 4,4 only
                 changes the printed MIP backbone (mipgen.cpp:200), not the hot path

Every generator is a pure function of its seed and of the region index, so any shard of a workload can be built without
building the rest (bench.py --regions N, multi-GPU sharding).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import capi, synth

CHROMS_24 = [str(i) for i in range(1, 23)] + ["X", "Y"]

# rho of the synthetic SVR model per workload, placed (rho_for_exit_rate below, measured on 800 exons of each config) so that ~12 % of the
# arm-sum lists' last pairs score above the reference's upper limit (2.2): the replay of the early exits (mipgen.cpp:430,434) then really
# removes candidates - practice62 / capture 140-180: 57 % of the dense grid, exome / 150-170: 19 %, exome + SNPs / 120-250: 56 %.  (With the
# practice62 value the exome's last pairs - l = 18, the lowest-scoring arm of the synthetic model - never reach 2.2: emitted == dense.)
MODEL_RHO = {"practice62": -2.2, "regions5k": -2.2, "exome": -4.568, "exome_snp": -1.645}


def practice62(seed: int = 20140101, genome_len: int = 400_000, n_regions: int = 62) -> Tuple[bytes, List[synth.Interval]]:
    """`practice62`: one 400 kb chromosome "7", 62 exon-like intervals (EGFR/TERT/BRAF-like)."""
    genome = synth.random_genome(genome_len, seed)
    ivs = synth.practice62_intervals("7", seed=seed, n=n_regions)
    # keep every interval inside the chromosome with room for the +/- (max_capture + 1000) context
    ivs = [iv for iv in ivs if iv.bed_end + 1500 < genome_len]
    return genome, ivs


def svr_model_path(cache_dir: str, genome: bytes, n_sv: int, seed: int = 7, gamma: Optional[float] = None,
                   coef_scale: float = 1.0, rho: Optional[float] = None) -> str:
    os.makedirs(cache_dir, exist_ok=True)
    tag = (f"svr_syn_{n_sv}_s{seed}" + (f"_g{gamma:g}" if gamma is not None else "") + (f"_c{coef_scale:g}" if coef_scale != 1.0 else "")
           + (f"_r{rho:g}" if rho is not None else ""))
    path = os.path.join(cache_dir, tag + ".model")
    if not os.path.exists(path):
        tmp = path + f".tmp{os.getpid()}"
        kw = {} if gamma is None else {"gamma": gamma}
        if rho is not None:
            kw["rho"] = rho
        synth.synthetic_svr_model(tmp, genome, n_sv, seed=seed, coef_scale=coef_scale, **kw)
        os.replace(tmp, path)
    return path


def build_regions(acc: Optional[capi.Accel], genome: bytes, ivs: List[synth.Interval], params: capi.Params,
                  bwa_mode: str = "unique", with_lrc: bool = True) -> List[capi.RegionData]:
    """Region records as the reference's -genome_dir input stage produces them; the 44 long-range k-mer frequencies
    come from the device kernel (mipgen_accel_long_range_content_batch) when an accelerator handle is given."""
    out = [capi.build_region(genome, iv.chrom, iv.bed_start, iv.bed_end, params, label=iv.label, bwa_mode=bwa_mode) for iv in ivs]
    if with_lrc and acc is not None:
        fill_long_range(acc, out, {ivs[0].chrom: genome} if ivs else {}, params)
    return out


def fill_long_range(acc: capi.Accel, regions: Sequence[capi.RegionData], genomes: Dict[str, bytes], params: capi.Params,
                    chunk: int = 4096) -> None:
    """Featurev5::get_long_range_content of every region (region +/- 1000 bases, /root/reference/mipgen.cpp:1225) in batched launches."""
    for lo in range(0, len(regions), chunk):
        part = regions[lo:lo + chunk]
        seqs, starts, stops = [], [], []
        for rd in part:
            g = genomes[rd.chrom]
            n = rd.c.seq_stop - rd.c.seq_start + 1
            s0 = max(0, rd.c.start_flanked - params.max_capture_size - 1 - 1000)
            seqs.append(g[s0:s0 + n + 2000])
            starts.append(rd.c.seq_start)
            stops.append(rd.c.seq_stop)
        lrc = acc.long_range_content_batch(seqs, starts, stops)
        for rd, v in zip(part, lrc):
            for i in range(capi.N_LRC):
                rd.c.long_range_content[i] = v[i]


# ----------------------------------------------------------------------------------------------------------------------
# configs[2]: 1,000 x 5 kb
# ----------------------------------------------------------------------------------------------------------------------

def regions5k_genome(seed: int = 3, length: int = 12_000_000 + 20_000) -> bytes:
    """Chromosome "1": iid ACGT, 0.1 % of the bases inside runs of 50 N (they trip the -1000 / all-zero guard)."""
    return synth.random_genome(length, seed, n_run_frac=0.001, n_run_len=50)


def regions5k_intervals(n: int = 1000, first: int = 0) -> List[synth.Interval]:
    return [synth.Interval("1", 5000 + i * 12_000, 5000 + i * 12_000 + 5000, f"reg{i + 1}") for i in range(first, first + n)]


def copy_table_5k(region_index: int, seq_len: int, lengths: Sequence[int], seed: int = 5) -> Dict[int, np.ndarray]:
    """Per-oligo copy numbers of one region: 98 % 1, 1.5 % uniform in 2..20, 0.5 % 101."""
    out: Dict[int, np.ndarray] = {}
    for ln in lengths:
        rng = np.random.default_rng([seed, region_index, ln])
        u = rng.random(seq_len)
        c = np.ones(seq_len, dtype=np.int32)
        mid = (u >= 0.98) & (u < 0.995)
        c[mid] = rng.integers(2, 21, size=int(mid.sum()), dtype=np.int32)
        c[u >= 0.995] = 101
        c[max(0, seq_len - ln):] = 0          # the reference never writes the oligos that would run past the region string (mipgen.cpp:829)
        out[ln] = c
    return out


def regions5k(n: int = 1000, first: int = 0, genome: Optional[bytes] = None) -> Tuple[bytes, List[synth.Interval]]:
    return (genome if genome is not None else regions5k_genome()), regions5k_intervals(n, first)


def build_regions5k(acc: Optional[capi.Accel], genome: bytes, ivs: Sequence[synth.Interval], params: capi.Params,
                    with_lrc: bool = True) -> List[capi.RegionData]:
    pairs = capi.arm_pairs_of(params)
    lengths = sorted({e for e, _ in pairs} | {l for _, l in pairs})
    out = []
    for iv in ivs:
        idx = (iv.bed_start - 5000) // 12_000
        rd = fast_region(genome, iv, params)
        copy = copy_table_5k(idx, rd.c.seq_len, lengths)
        rd = capi.RegionData(rd.c.start_flanked, rd.c.stop_flanked, rd.c.seq_start, rd.seq, copy=copy, chrom=iv.chrom, label=iv.label,
                             start=iv.bed_start + 1, stop=iv.bed_end)
        out.append(rd)
    if with_lrc and acc is not None:
        fill_long_range(acc, out, {"1": genome}, params)
    return out


# ----------------------------------------------------------------------------------------------------------------------
# configs[3]/[4]: exome200k
# ----------------------------------------------------------------------------------------------------------------------

def exome_layout(total_bp: int = 300_000_000, n_intervals: int = 200_000, seed: int = 11) -> Tuple[Dict[str, int], List[synth.Interval]]:
    """Chromosome lengths (24 chromosomes, decreasing sizes) and the 200,000 sorted, non-overlapping intervals
    (the order the reference's sort produces: chromosome as a string, then start; mipgen.cpp:37-67,999)."""
    rng = np.random.default_rng(seed)
    w = np.linspace(2.0, 0.5, len(CHROMS_24))
    lens = np.floor(w / w.sum() * total_bp).astype(np.int64)
    chrom_len = {c: int(l) for c, l in zip(CHROMS_24, lens)}
    counts = np.floor(lens / lens.sum() * n_intervals).astype(np.int64)
    counts[0] += n_intervals - counts.sum()
    ivs: List[synth.Interval] = []
    for c, L, k in sorted(zip(CHROMS_24, lens, counts), key=lambda t: t[0]):
        ln = np.clip(np.round(np.exp(rng.normal(np.log(130.0), 0.8, size=int(k)))), 20, 10_000).astype(np.int64)
        # gaps: at least 200 bases (the reference merges intervals closer than min_capture / 2, mipgen.cpp:1019), the rest spread evenly at random
        room = int(L) - 6000 - int(ln.sum()) - 200 * int(k)
        cuts = np.sort(rng.integers(0, max(room, 1), size=int(k)))
        gaps = np.diff(np.concatenate([[0], cuts])) + 200
        starts = 3000 + np.cumsum(gaps) + np.concatenate([[0], np.cumsum(ln)[:-1]])
        for i in range(int(k)):
            ivs.append(synth.Interval(c, int(starts[i]), int(starts[i] + ln[i]), f"ex{c}_{i + 1}"))
    return chrom_len, ivs


_chrom_cache: Dict[Tuple[str, int], bytes] = {}


def exome_chromosome(chrom: str, length: int, seed: int = 11) -> bytes:
    key = (chrom, seed)
    if key not in _chrom_cache:
        _chrom_cache[key] = synth.random_genome(length, [seed, CHROMS_24.index(chrom)], gc=0.41)
    return _chrom_cache[key]


def exome_snp_class(chrom: str, length: int, genome: bytes, seed: int = 13) -> np.ndarray:
    """Per-base SNP class of a chromosome (0 none / 1 alternate-allele arm can be generated / 2 it cannot), from a synthetic table
    of 1 SNP per 300 bp: biallelic SNVs whose reference allele matches the genome (class 1) and 5 % multi-base references, whose
    2nd..last bases carry an allele string longer than two characters (class 2; mipgen.cpp:644,959-965)."""
    rng = np.random.default_rng([seed, CHROMS_24.index(chrom)])
    n = length // 300
    pos = np.unique(rng.integers(1000, length - 1000, size=n))
    cls = np.zeros(length, dtype=np.uint8)
    indel = rng.random(pos.size) < 0.05
    cls[pos[~indel]] = 1
    ref_len = rng.integers(2, 5, size=int(indel.sum()))
    for p, k in zip(pos[indel], ref_len):
        cls[p + 1:p + int(k)] = 2
    g = np.frombuffer(genome, dtype=np.uint8)
    cls[g == ord("N")] = 0
    return cls


def fast_region(genome: bytes, iv: synth.Interval, params: capi.Params, snp_class: Optional[np.ndarray] = None, pad: int = 15) -> capi.RegionData:
    """A region as the -genome_dir input stage slices it (mipgen.cpp:1214-1220) with every oligo copy = 1 and no unmappable
    windows (copy table NULL): the cheap form for workloads of 10^4-10^5 regions."""
    maxC = params.max_capture_size
    sf, ef = iv.bed_start + 1, iv.bed_end
    cs = max(1, sf - maxC)
    ce = min(len(genome), ef + maxC + pad)
    sc = snp_class[cs - 1:ce] if snp_class is not None else None
    return capi.RegionData(sf, ef, cs, genome[cs - 1:ce], snp_class=sc, chrom=iv.chrom, label=iv.label, start=sf, stop=ef)


def build_exome(acc: Optional[capi.Accel], chrom_len: Dict[str, int], ivs: Sequence[synth.Interval], params: capi.Params,
                snps: bool = False, with_lrc: bool = True) -> List[capi.RegionData]:
    out = []
    genomes: Dict[str, bytes] = {}
    snp: Dict[str, np.ndarray] = {}
    for iv in ivs:
        if iv.chrom not in genomes:
            genomes[iv.chrom] = exome_chromosome(iv.chrom, chrom_len[iv.chrom])
            if snps:
                snp[iv.chrom] = exome_snp_class(iv.chrom, chrom_len[iv.chrom], genomes[iv.chrom])
        out.append(fast_region(genomes[iv.chrom], iv, params, snp.get(iv.chrom)))
    if with_lrc and acc is not None:
        fill_long_range(acc, out, genomes, params)
    return out


def shard_weights(ivs: Sequence[synth.Interval], params: capi.Params, svr: bool) -> np.ndarray:
    """Relative device time of every interval (mipgen_amd.dist.region_cost): what contiguous region ranges are balanced by."""
    from . import dist as mdist
    pairs = capi.arm_pairs_of(params)
    sums = [e + l for e, l in pairs]
    cand, n_pos, n_sizes = dense_candidates(ivs, params, detail=True)
    return mdist.region_cost(cand, n_pos, n_sizes, len({e for e, _ in pairs}), len({l for _, l in pairs}), params.capture_increment, max(sums) - min(sums), svr)


def dense_candidates(ivs: Sequence[synth.Interval], params: capi.Params, detail: bool = False):
    """Dense-grid size of every interval: mipgen.cpp:421-429 (detail: also the scan positions and surviving capture sizes)."""
    pairs = capi.arm_pairs_of(params)
    max_sum = max(e + l for e, l in pairs)
    K = capi.n_sizes_all(params)
    out = np.empty(len(ivs), dtype=np.int64)
    npos = np.empty(len(ivs), dtype=np.int64)
    nk = np.empty(len(ivs), dtype=np.int64)
    for i, iv in enumerate(ivs):
        sf, ef = iv.bed_start + 1, iv.bed_end
        cur = max(0, sf - params.max_capture_size + max_sum)
        k0 = 0
        while k0 < K:
            C = params.max_capture_size - k0 * params.capture_increment
            if C > ef - sf + params.max_mip_overlap and C - params.capture_increment >= params.min_capture_size:
                k0 += 1
            else:
                break
        out[i] = max(0, ef - cur) * (K - k0) * len(pairs) * 2
        npos[i] = max(0, ef - cur)
        nk[i] = K - k0
    return (out, npos, nk) if detail else out


def rho_for_exit_rate(P: capi.Params, grids, scores: np.ndarray, rho_now: float, target: float = 0.12) -> float:
    """The rho that puts `target` of the arm-sum lists' LAST pairs (max over the two strands: previous_best_score after the list,
    /root/reference/mipgen.cpp:495) above the upper score limit, given dense scores computed with rho_now - rho is an additive constant of
    the SVR score (svm.cpp:2515), so the early exits of mipgen.cpp:430,434 can be placed on any workload without re-scoring."""
    A = P.n_arm_pairs
    sums = [P.arm_ext[i] + P.arm_lig[i] for i in range(A)]
    last = [i for i in range(A) if i == A - 1 or sums[i + 1] != sums[i]]
    vals = []
    for g in grids:
        if g.count == 0:
            continue
        s = scores[g.offset:g.offset + g.count].reshape(g.n_pos, g.n_sizes, 2, A)[..., last]
        with np.errstate(invalid="ignore"):
            v = np.nanmax(s, axis=2).ravel()
        vals.append(v[np.isfinite(v) & (v != 0.0)])
    v = np.concatenate(vals)
    q = float(np.quantile(v, 1.0 - target))
    return round(q + rho_now - P.upper_score_limit, 3)

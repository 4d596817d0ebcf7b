"""Seeded synthetic inputs for the MIPgen hot path (genomes, BED intervals, copy/mappability/SNP/mask
tables, libsvm models).

No real genome, BED, VCF or trained model is available offline, so every workload
named in BASELINE.json is replaced by a synthetic stand-in of the same shape, generated here from fixed
seeds.  The `shim_*` functions reproduce, bit for bit, the deterministic rules implemented by the
external-tool stand-ins under oracle/ (fakebwa.sh, faketrf.sh) so that the tables handed to the
accelerator are the very tables the reference binary derives from those stand-ins.

Nothing in this file touches the GPU or the oracle.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

# ----------------------------------------------------------------------------------------------
# genomes and intervals
# ----------------------------------------------------------------------------------------------

_BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


def random_genome(length: int, seed: int, gc: float = 0.5, n_run_frac: float = 0.0, n_run_len: int = 50) -> bytes:
    """iid ACGT chromosome; optionally a fraction of bases replaced by runs of N."""
    rng = np.random.default_rng(seed)
    p = np.array([(1 - gc) / 2, gc / 2, gc / 2, (1 - gc) / 2])
    seq = _BASES[rng.choice(4, size=length, p=p)].copy()
    if n_run_frac > 0:
        n_runs = max(1, int(length * n_run_frac / n_run_len))
        starts = rng.integers(0, max(1, length - n_run_len), size=n_runs)
        for s in starts:
            seq[s:s + n_run_len] = ord("N")
    return seq.tobytes()


# zones of hard_genome (0-based half-open; the designs of tests/golden/make_golden.py: DESIGNS4 lay their intervals over them)
HARD_ZONES = {"iupac": (4000, 9000), "lower": (10000, 14000), "lowcomplex": (16000, 22000), "gc20": (24000, 26000), "gc70": (28000, 30000),
              "dash": (32000, 34000)}


def hard_genome(length: int = 40000, seed: int = 404) -> bytes:
    """A chromosome AS IT STANDS IN A FASTA FILE with everything the iid ACGT genomes lack (GRCh37 has all of it):
      iupac       ambiguity codes R Y M K S W B D H V, one per ~120 bases (the reference guards on N only, SVMipv4.cpp:116; any other byte just is
                  not A/C/G/T to its counters, :118-141, and passes through reverse_comp unchanged, MinusSVMipv4.cpp:24-25)
      lower       soft-masked stretches of 40-300 lower-case bases, ambiguity codes and n included (the reference upper-cases what it reads,
                  mipgen.cpp:1158,1168,1208)
      lowcomplex  homopolymers of 30-70 bases, (CA)n / (AT)n / (GGC)n / (AAAG)n microsatellites of 40-140 bases, one every ~350 bases
      gc20, gc70  blocks of 20 % and 70 % G+C
      dash        '-' bytes, one per ~400 bases (a '-' in an arm is the other half of the guard of SVMipv4.cpp:116)
    with the usual iid background and short runs of N everywhere."""
    rng = np.random.default_rng(seed)
    seq = np.frombuffer(random_genome(length, seed + 1, n_run_frac=0.002, n_run_len=7), dtype=np.uint8).copy()
    lo, hi = HARD_ZONES["iupac"]
    codes = np.frombuffer(b"RYMKSWBDHV", dtype=np.uint8)
    for p in rng.choice(np.arange(lo, hi), size=(hi - lo) // 120, replace=False):
        seq[p] = codes[rng.integers(0, len(codes))]
    lo, hi = HARD_ZONES["gc20"]
    seq[lo:hi] = np.frombuffer(random_genome(hi - lo, seed + 2, gc=0.2), dtype=np.uint8)
    lo, hi = HARD_ZONES["gc70"]
    seq[lo:hi] = np.frombuffer(random_genome(hi - lo, seed + 3, gc=0.7), dtype=np.uint8)
    lo, hi = HARD_ZONES["lowcomplex"]
    p = lo + 150
    units = [b"A", b"T", b"G", b"C", b"CA", b"AT", b"GGC", b"AAAG", b"TG", b"CCG"]
    k = 0
    while p < hi - 200:
        u = units[k % len(units)]
        n = int(rng.integers(30, 71)) if len(u) == 1 else int(rng.integers(40, 141))
        seq[p:p + n] = np.frombuffer((u * (n // len(u) + 1))[:n], dtype=np.uint8)
        p += n + int(rng.integers(200, 420))
        k += 1
    lo, hi = HARD_ZONES["dash"]
    for p in rng.choice(np.arange(lo, hi), size=(hi - lo) // 400, replace=False):
        seq[p] = ord("-")
    lo, hi = HARD_ZONES["lower"]
    for p in rng.choice(np.arange(lo, hi), size=(hi - lo) // 200, replace=False):      # ambiguity codes here too (they come out of toupper as upper case)
        seq[p] = codes[rng.integers(0, len(codes))]
    p = lo + 60
    while p < hi - 320:
        n = int(rng.integers(40, 301))
        seq[p:p + n] |= 0x20                                  # ASCII lower case
        p += n + int(rng.integers(80, 500))
    return seq.tobytes()


def write_fasta(path: str, name: str, seq: bytes, width: int = 60) -> None:
    with open(path, "wb") as fh:
        fh.write(b">" + name.encode() + b"\n")
        for i in range(0, len(seq), width):
            fh.write(seq[i:i + width] + b"\n")


@dataclass
class Interval:
    chrom: str
    bed_start: int  # 0-based, BED
    bed_end: int    # exclusive
    label: str = ""


def practice62_intervals(chrom: str = "7", seed: int = 20140101, first_start: int = 5000,
                         n: int = 62, min_len: int = 60, max_len: int = 400, min_gap: int = 2000) -> List[Interval]:
    """62 exon-like intervals (28+16+18, mimicking EGFR/TERT/BRAF coding exons), SURVEY.md section 8d."""
    rng = np.random.default_rng(seed)
    genes = ["EGFR"] * 28 + ["TERT"] * 16 + ["BRAF"] * 18
    genes = (genes * ((n + 61) // 62))[:n]
    out: List[Interval] = []
    pos = first_start
    for i in range(n):
        ln = int(round(np.exp(rng.uniform(np.log(min_len), np.log(max_len)))))
        out.append(Interval(chrom, pos, pos + ln, f"{genes[i]}_ex{i + 1}"))
        pos += ln + min_gap + int(rng.integers(0, 3000))
    return out


def tiled_intervals(chrom: str, n: int, length: int, stride: int, first_start: int = 5000, label: str = "reg") -> List[Interval]:
    return [Interval(chrom, first_start + i * stride, first_start + i * stride + length, f"{label}{i + 1}") for i in range(n)]


def write_bed(path: str, intervals: Sequence[Interval], with_chr_prefix: bool = True) -> None:
    with open(path, "w") as fh:
        for iv in intervals:
            c = ("chr" + iv.chrom) if with_chr_prefix else iv.chrom
            if iv.label:
                fh.write(f"{c}\t{iv.bed_start}\t{iv.bed_end}\t{iv.label}\n")
            else:
                fh.write(f"{c}\t{iv.bed_start}\t{iv.bed_end}\n")


# ----------------------------------------------------------------------------------------------
# deterministic rules of the external-tool stand-ins (oracle/fakebwa.sh, oracle/faketrf.sh)
# ----------------------------------------------------------------------------------------------

def shim_copy(start: np.ndarray, length: int, mode: str = "unique") -> np.ndarray:
    """Copy number the fakebwa stand-in reports for the arm oligo [start, start+length-1].

    Mirrors the awk in oracle/fakebwa.sh; a read with no X0 tag becomes copy 100 in the reference
    (/root/reference/mipgen.cpp:589-592)."""
    start = np.asarray(start, dtype=np.int64)
    if mode == "unique":
        return np.ones_like(start, dtype=np.int32)
    if mode == "blocks":
        return np.where(((start % 3000) >= 1200) & ((start % 3000) < 1500), 500, 1).astype(np.int32)
    h = (start * 7919 + length * 104729) % 1000
    copy = np.ones_like(start, dtype=np.int64)
    copy = np.where((h >= 940) & (h < 970), 2 + (h % 19), copy)
    copy = np.where((h >= 970) & (h < 985), 21 + (h % 60), copy)
    copy = np.where((h >= 985) & (h < 995), 101 + (h % 400), copy)
    copy = np.where((h >= 995) & (h < 998), 100, copy)   # untagged read
    copy = np.where(h >= 998, 0, copy)
    return copy.astype(np.int32)


def shim_unmappable(pos: np.ndarray, size: int, mode: str = "unique") -> np.ndarray:
    """True where the fakebwa stand-in reports the capture window (size, pos) as non-unique."""
    pos = np.asarray(pos, dtype=np.int64)
    if mode in ("unique", "blocks"):
        return np.zeros(pos.shape, dtype=bool)
    return ((pos * 31 + size * 17) % 211) == 0


def shim_mask(seq: bytes, record_index: int) -> bytes:
    """Masked copy of a region sequence as produced by oracle/faketrf.sh for FASTA record `record_index`."""
    a = np.frombuffer(seq, dtype=np.uint8).copy()
    off = np.arange(len(a), dtype=np.int64)
    hit = (((off // 8) * 131 + record_index * 17) % 23) == 0
    a[hit] = ord("N")
    return a.tobytes()


# ----------------------------------------------------------------------------------------------
# SNPs
# ----------------------------------------------------------------------------------------------

@dataclass
class Snp:
    chrom: str
    pos: int       # 1-based
    ref: str
    alt: str


def random_snps(chrom: str, genome: bytes, lo: int, hi: int, seed: int, per_bp: float = 1 / 300.0,
                indel_frac: float = 0.05, odd_frac: float = 0.05) -> List[Snp]:
    """Biallelic SNVs in [lo, hi] (1-based), a few multi-base refs (indel style, /root/reference/mipgen.cpp:959-965)
    and a few records whose ref allele does not match the genome on either strand (alt-MIP generation fails)."""
    rng = np.random.default_rng(seed)
    n = max(1, int((hi - lo + 1) * per_bp))
    pos = np.unique(rng.integers(lo, hi + 1, size=n))
    out: List[Snp] = []
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    for p in pos:
        g = chr(genome[p - 1])
        if g not in comp:
            continue
        u = rng.random()
        others = [b for b in "ACGT" if b != g]
        alt = others[int(rng.integers(0, 3))]
        if u < indel_frac:
            ln = int(rng.integers(2, 5))
            ref = genome[p - 1:p - 1 + ln].decode()
            out.append(Snp(chrom, int(p), ref, g))
        elif u < indel_frac + odd_frac:
            # ref allele that matches neither the base nor its complement
            bad = [b for b in "ACGT" if b != g and b != comp[g]]
            out.append(Snp(chrom, int(p), bad[0], alt))
        elif u < indel_frac + odd_frac + 0.25:
            # allele reported on the minus strand
            out.append(Snp(chrom, int(p), comp[g], comp[alt]))
        else:
            out.append(Snp(chrom, int(p), g, alt))
    return out


def write_vcf(path: str, snps: Sequence[Snp]) -> None:
    with open(path, "w") as fh:
        fh.write("##fileformat=VCFv4.1\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n")
        for i, s in enumerate(snps):
            fh.write(f"{s.chrom}\t{s.pos}\trs{i + 1}\t{s.ref}\t{s.alt}\t.\t.\t.\n")


def snp_table(snps: Sequence[Snp]) -> Dict[str, Dict[int, str]]:
    """chr -> pos -> ref+alt, as parse_vcf builds it (/root/reference/mipgen.cpp:933-978)."""
    tab: Dict[str, Dict[int, str]] = {}
    for s in snps:
        d = tab.setdefault(s.chrom, {})
        alleles = s.ref + s.alt
        if len(s.ref) > 1:
            for i in range(1, len(s.ref)):
                d[s.pos + i] = alleles
        else:
            d[s.pos] = alleles
    return tab


# ----------------------------------------------------------------------------------------------
# default arm-length pairs (reference default: sums 40..45, ext>=16, lig>=18, both <=30)
# ----------------------------------------------------------------------------------------------

def arm_pairs_from_sums(sums: Sequence[int] = (40, 41, 42, 43, 44, 45), ext_min: int = 16, lig_min: int = 18) -> List[Tuple[int, int]]:
    """(ext, lig) pairs in enumeration order: sum descending, ext ascending (/root/reference/mipgen.cpp:237-261,431-438)."""
    by_sum: Dict[int, List[int]] = {}
    for s in sums:
        lst = by_sum.setdefault(s, [])
        lig = lig_min
        while lig <= s - ext_min and lig <= 30:
            e = s - lig
            if e <= 30:
                lst.append(e)
            lig += 1
        lst.sort()
    out: List[Tuple[int, int]] = []
    for s in sorted(by_sum, reverse=True):
        for e in by_sum[s]:
            out.append((e, s - e))
    return out


# ----------------------------------------------------------------------------------------------
# synthetic libsvm models
# ----------------------------------------------------------------------------------------------

_ARM_MERS = ["A", "AA", "AC", "AG", "AT", "C", "CA", "CC", "CG", "CT", "G", "GA", "GC", "GG", "GT", "T", "TA", "TC", "TG", "TT"]


def _all_mers(kmax: int) -> List[str]:
    out: List[str] = []

    def rec(prefix: str) -> None:
        if prefix:
            out.append(prefix)
        if len(prefix) < kmax:
            for b in "ACGT":
                rec(prefix + b)
    rec("")
    return out


_INSERT_MERS = _all_mers(3)          # lexicographic, "A","AA","AAA",... (84 entries)
_JUNCTIONS = [a + b for a in "ACGT" for b in "ACGT"]


def _count_overlapping(s: str, sub: str) -> int:
    n = 0
    i = s.find(sub)
    while i != -1:
        n += 1
        i = s.find(sub, i + 1)
    return n


def feature_vector_py(ext: str, lig: str, insert: str, ext_copy: int, lig_copy: int, lrc: Sequence[float]) -> np.ndarray:
    """Plain-Python 192-feature builder used ONLY to draw synthetic support vectors near the data
    distribution (layout SURVEY.md Appendix A.4).  Not a parity reference."""
    if "N" in ext or "N" in lig:
        return np.zeros(192)
    v: List[float] = []
    for arm in (ext,):
        for m in _ARM_MERS:
            if m == "T":
                v.append((arm.count("G") + arm.count("C")) / len(arm))
            v.append(_count_overlapping(arm, m) / (len(arm) - len(m) + 1.0))
        v.append(float(len(arm)))
    v.extend(float(x) for x in lrc)
    for m in _INSERT_MERS:
        if m == "T":
            v.append((insert.count("G") + insert.count("C")) / len(insert))
        v.append(_count_overlapping(insert, m) / (len(insert) - len(m) + 1.0))
    v.append(float(len(insert)))
    for m in _ARM_MERS:
        if m == "T":
            v.append((lig.count("G") + lig.count("C")) / len(lig))
        v.append(_count_overlapping(lig, m) / (len(lig) - len(m) + 1.0))
    v.append(float(len(lig)))
    lj = lig[:2]
    v.extend(1.0 if lj == j else 0.0 for j in _JUNCTIONS)
    v.append(2.0 if ext_copy > 100 else (np.log10(ext_copy) if ext_copy > 0 else -np.inf))
    v.append(2.0 if lig_copy > 100 else (np.log10(lig_copy) if lig_copy > 0 else -np.inf))
    assert len(v) == 192
    return np.array(v)


_COMP = bytes.maketrans(b"ACGT", b"TGCA")


def revcomp(s: str) -> str:
    return s.encode().translate(_COMP)[::-1].decode()


def synthetic_svr_model(path: str, genome: bytes, n_sv: int, seed: int = 7, gamma: float = 0.05 / 192.0 * 40.0,
                        rho: float = -1.5, drop_zero_frac: float = 0.2, capture: Tuple[int, int] = (140, 180),
                        coef_scale: float = 1.0) -> None:
    """Write a libsvm 3.17 text model (epsilon_svr / rbf) whose support vectors are feature vectors of
    random real candidates drawn from `genome`, so that RBF distances are O(1) and the kernel values
    spread over (0,1) (SURVEY.md sections 7.3, 8d).  Values use %.8g as svm_save_model does
    (/root/reference/svm.cpp:2725); a fraction of exactly-zero entries is omitted to exercise the sparse form."""
    rng = np.random.default_rng(seed)
    pairs = arm_pairs_from_sums()
    g = genome
    L = len(g)
    lrc = np.full(44, 0.0)
    with open(path, "w") as fh:
        fh.write("svm_type epsilon_svr\nkernel_type rbf\n")
        fh.write("gamma %g\n" % gamma)
        fh.write("nr_class 2\ntotal_sv %d\nrho %g\nSV\n" % (n_sv, rho))
        made = 0
        while made < n_sv:
            e, l = pairs[int(rng.integers(0, len(pairs)))]
            C = int(rng.integers(capture[0], capture[1] + 1))
            p = int(rng.integers(1000, L - 1000))
            ss = C - e - l
            minus = bool(rng.integers(0, 2))
            ins = g[p - 1:p - 1 + ss].decode()
            if not minus:
                ext = g[p - 1 - e:p - 1].decode()
                lig = g[p - 1 + ss:p - 1 + ss + l].decode()
            else:
                ext = revcomp(g[p - 1 + ss:p - 1 + ss + e].decode())
                lig = revcomp(g[p - 1 - l:p - 1].decode())
                ins = revcomp(ins)
            if "N" in ext or "N" in lig:
                continue
            # long-range content: plausible frequencies (the real ones are per-region constants)
            for k in range(44):
                lrc[k] = rng.uniform(0.0, 0.3)
            ec = 1 if rng.random() < 0.9 else int(rng.integers(2, 150))
            lc = 1 if rng.random() < 0.9 else int(rng.integers(2, 150))
            x = feature_vector_py(ext, lig, ins, ec, lc, lrc)
            coef = rng.uniform(-1.0, 1.0) * coef_scale
            parts = ["%.16g" % coef]
            for j, val in enumerate(x):
                if val == 0.0 and rng.random() < drop_zero_frac:
                    continue
                parts.append("%d:%.8g" % (j + 1, val))
            fh.write(" ".join(parts) + " \n")
            made += 1

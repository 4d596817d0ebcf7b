#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REAL reference (oracle/_ref, compiled by
oracle/Makefile from the sources under /root/reference).  Run in the build container only:

    make -C oracle all && python tests/golden/make_golden.py

What is committed is DATA: inputs (small synthetic genomes / BEDs / VCFs / libsvm models, all generated here
from fixed seeds) and the reference's outputs on them.  No reference source text is stored.

Fixtures
  candidates.npz / candidates.json   per-candidate known answers from the reference classes
                                     (SVMipv4::get_score, SVMipv4::get_parameters, svm_predict,
                                      Featurev5::get_long_range_content) at full double precision
  design_<name>/                      end-to-end designs run through the reference binary: inputs + the
                                      reference's all_mips (gz) / collapsed / picked / snp files + sha256
"""
from __future__ import annotations

import ctypes as C
import gzip
import hashlib
import json
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from mipgen_amd import synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from oracle.run_reference import run_reference  # noqa: E402

MIDDLE = b"CTTCAGCTTCCCGATATCCGACGGTAGTGTNNNNN"   # universal_middle_mip_seq for the default -tag_sizes 5,0


def sha256(path: str) -> str:
    h = hashlib.sha256()
    with open(path, "rb") as fh:
        for blk in iter(lambda: fh.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def gen_models(genome: bytes) -> None:
    os.makedirs(os.path.join(HERE, "models"), exist_ok=True)
    synth.synthetic_svr_model(os.path.join(HERE, "models", "svr_syn_64.model"), genome, 64, seed=7)
    synth.synthetic_svr_model(os.path.join(HERE, "models", "svr_syn_200.model"), genome, 200, seed=8, drop_zero_frac=0.5)
    gen_short_model(genome)


def gen_short_model(genome: bytes) -> None:
    # support vectors from SHORT captures (scan sizes 2..70): the model of the designs that leave the tiled SVR kernel's limits
    synth.synthetic_svr_model(os.path.join(HERE, "models", "svr_syn_short_48.model"), genome, 48, seed=9, gamma=0.004, rho=1.535, capture=(47, 110))


def gen_candidates(genome: bytes) -> None:
    """Known answers at full precision, incl. edge cases: N in arm / insert, copy 0/1/100/101, both strands,
    every junction, scan sizes past the 250 clamp."""
    R = po.refdrv()
    rng = np.random.default_rng(42)
    dp = C.POINTER(C.c_double)
    m64 = R.ref_svm_load_model(os.path.join(HERE, "models", "svr_syn_64.model").encode())
    m200 = R.ref_svm_load_model(os.path.join(HERE, "models", "svr_syn_200.model").encode())
    g = genome
    recs = []
    logistic, params, svr64, svr200, lrcs = [], [], [], [], []
    n = 400
    for t in range(n):
        e = int(rng.integers(16, 31)); l = int(rng.integers(16, 31))
        ss = int(rng.integers(70, 230)) if t % 10 else int(rng.integers(250, 300))
        p = int(rng.integers(400, len(g) - 700)); strand = int(rng.integers(0, 2))
        ext = bytearray(g[p - 1 - e:p - 1] if strand == 0 else g[p - 1 + ss:p - 1 + ss + e])
        ins = bytearray(g[p - 1:p - 1 + ss])
        lig = bytearray(g[p - 1 + ss:p - 1 + ss + l] if strand == 0 else g[p - 1 - l:p - 1])
        kind = t % 20
        if kind == 3: ext[int(rng.integers(0, e))] = ord("N")
        if kind == 5: lig[int(rng.integers(0, l))] = ord("N")
        if kind == 7:
            a = int(rng.integers(0, ss - 12)); ins[a:a + int(rng.integers(1, 12))] = b"N" * 12
            ins = ins[:ss]
        if kind == 9: ins[0] = ord("N")
        if kind == 11: ins[int(rng.integers(0, ss))] = ord("R")          # IUPAC code: not N, not ACGT
        if kind == 13: lig[0 if strand == 0 else l - 1] = ord("N")       # junction with N -> guard
        ec = int(rng.choice([0, 1, 1, 1, 1, 2, 7, 20, 21, 75, 100, 101, 500, 70000]))
        lc = int(rng.choice([0, 1, 1, 1, 1, 3, 19, 100, 101, 1000]))
        lrc = rng.uniform(0, 0.3, 44)
        ext, lig, ins = bytes(ext), bytes(lig), bytes(ins)
        s = R.ref_logistic(strand, ext, lig, ins, ec, lc, MIDDLE)
        x = np.empty(192)
        R.ref_parameters(strand, ext, lig, ins, ec, lc, MIDDLE, lrc.ctypes.data_as(dp), x.ctypes.data_as(dp))
        recs.append({"strand": strand, "ext_fwd": ext.decode(), "lig_fwd": lig.decode(), "ins_fwd": ins.decode(),
                     "ext_copy": ec, "lig_copy": lc})
        logistic.append(s); params.append(x); lrcs.append(lrc)
        svr64.append(R.ref_predict_text(m64, x.ctypes.data_as(dp), 192))
        svr200.append(R.ref_predict_text(m200, x.ctypes.data_as(dp), 192))
    # long-range content known answers
    lr_in, lr_out = [], []
    for t in range(6):
        a = int(rng.integers(0, len(g) - 4000)); ln = int(rng.integers(2300, 3500))
        seq = g[a:a + ln]
        cs = a + 1001; ce = a + ln - 1000
        out = np.empty(44)
        R.ref_long_range_content(seq, cs, ce, out.ctypes.data_as(dp))
        lr_in.append({"offset": a, "len": ln, "chrom_seq_start": cs, "chrom_seq_stop": ce}); lr_out.append(out)
    np.savez_compressed(os.path.join(HERE, "candidates.npz"), logistic=np.array(logistic), params=np.array(params),
                        svr64=np.array(svr64), svr200=np.array(svr200), lrc=np.array(lrcs), lr_out=np.array(lr_out))
    with open(os.path.join(HERE, "candidates.json"), "w") as fh:
        json.dump({"middle": MIDDLE.decode(), "candidates": recs, "long_range": lr_in}, fh)
    print(f"candidates: {n} known answers, {len(lr_in)} long-range vectors")


def gen_candidates_hard(genome4: bytes) -> None:
    """Known answers on sequence that is NOT iid ACGT + N: candidates cut from the hard genome (mipgen_amd/synth.py: hard_genome - ambiguity
    codes, '-' bytes, homopolymers, microsatellites, 20 % / 70 % GC), upper-cased as the reference's input stage hands them to its classes
    (mipgen.cpp:1158) and - every fifth - left in LOWER case (what the classes do with bytes their input stage would never pass: nothing matches
    'A'/'C'/'G'/'T', SVMipv4.cpp:118-141; reverse_comp leaves them, MinusSVMipv4.cpp:24-25).  Ambiguity codes and '-' in arms AND inserts."""
    R = po.refdrv()
    rng = np.random.default_rng(4321)
    dp = C.POINTER(C.c_double)
    m64 = R.ref_svm_load_model(os.path.join(HERE, "models", "svr_syn_64.model").encode())
    m200 = R.ref_svm_load_model(os.path.join(HERE, "models", "svr_syn_200.model").encode())
    zones = list(synth.HARD_ZONES.items())
    recs, logistic, params, svr64, svr200, lrcs = [], [], [], [], [], []
    n = 240
    for t in range(n):
        zname, (lo, hi) = zones[t % len(zones)]
        e = int(rng.integers(16, 31)); l = int(rng.integers(16, 31))
        ss = int(rng.integers(70, 200))
        p = int(rng.integers(lo + 40, hi - 260)); strand = int(rng.integers(0, 2))
        g = genome4 if t % 5 == 4 else genome4.upper()
        ext = bytearray(g[p - 1 - e:p - 1] if strand == 0 else g[p - 1 + ss:p - 1 + ss + e])
        ins = bytearray(g[p - 1:p - 1 + ss])
        lig = bytearray(g[p - 1 + ss:p - 1 + ss + l] if strand == 0 else g[p - 1 - l:p - 1])
        kind = t % 12
        codes = b"RYMKSWBDHV"
        if kind == 1: ext[int(rng.integers(0, e))] = codes[t % 10]               # an ambiguity code in the extension arm
        if kind == 2: lig[int(rng.integers(0, l))] = codes[(t + 3) % 10]
        if kind == 3: lig[0 if strand == 0 else l - 1] = codes[t % 10]           # ... at the ligation junction (no junction feature matches)
        if kind == 4: ext[int(rng.integers(0, e))] = ord("-")                    # '-' in an arm: the guard of SVMipv4.cpp:116
        if kind == 5: ins[int(rng.integers(0, ss))] = ord("-")                   # '-' in the insert only: no guard
        if kind == 6: ins[0] = codes[t % 10]; ins[ss - 1] = codes[(t + 5) % 10]
        if kind == 7: ext[e - 1] = codes[t % 10]; lig[l - 1] = codes[(t + 1) % 10]
        ec = int(rng.choice([0, 1, 1, 1, 2, 20, 100, 101])); lc = int(rng.choice([1, 1, 1, 3, 19, 101]))
        lrc = rng.uniform(0, 0.3, 44)
        ext, lig, ins = bytes(ext), bytes(lig), bytes(ins)
        s = R.ref_logistic(strand, ext, lig, ins, ec, lc, MIDDLE)
        x = np.empty(192)
        R.ref_parameters(strand, ext, lig, ins, ec, lc, MIDDLE, lrc.ctypes.data_as(dp), x.ctypes.data_as(dp))
        recs.append({"strand": strand, "ext_fwd": ext.decode(), "lig_fwd": lig.decode(), "ins_fwd": ins.decode(), "ext_copy": ec, "lig_copy": lc, "zone": zname})
        logistic.append(s); params.append(x); lrcs.append(lrc)
        svr64.append(R.ref_predict_text(m64, x.ctypes.data_as(dp), 192))
        svr200.append(R.ref_predict_text(m200, x.ctypes.data_as(dp), 192))
    # long-range content (Featurev5.cpp:18-56) over every zone, on the upper-cased chromosome (mipgen.cpp:1208) and once on the bytes as they stand
    lr_in, lr_out = [], []
    for t, (zname, (lo, hi)) in enumerate(zones + [("lower_raw", synth.HARD_ZONES["lower"])]):
        a = max(0, lo - 900); ln = min(len(genome4) - a, hi - lo + 1800)
        seq = (genome4 if zname == "lower_raw" else genome4.upper())[a:a + ln]
        cs = a + 1001; ce = a + ln - 1000
        out = np.empty(44)
        R.ref_long_range_content(seq, cs, ce, out.ctypes.data_as(dp))
        lr_in.append({"offset": a, "len": ln, "chrom_seq_start": cs, "chrom_seq_stop": ce, "raw": zname == "lower_raw"}); lr_out.append(out)
    np.savez_compressed(os.path.join(HERE, "candidates_hard.npz"), logistic=np.array(logistic), params=np.array(params),
                        svr64=np.array(svr64), svr200=np.array(svr200), lrc=np.array(lrcs), lr_out=np.array(lr_out))
    with open(os.path.join(HERE, "candidates_hard.json"), "w") as fh:
        json.dump({"middle": MIDDLE.decode(), "candidates": recs, "long_range": lr_in}, fh)
    print(f"candidates_hard: {n} known answers; guard (-1000) on {sum(1 for v in logistic if v == -1000.0)}")


def gen_libsvm_trained_model(genome: bytes) -> None:
    """A model TRAINED and WRITTEN by the reference's own libsvm (svm_train, svm.cpp:2095; svm_save_model, svm.cpp:2644-2757) on the feature vectors
    of 360 candidates of the golden genome (SVMipv4::get_parameters) with a smooth synthetic target, and the reference's predictions (svm_predict through
    the text hop of mipgen.cpp:1948-2019) for 150 other candidates, edge cases included.  Every other model fixture is written by mipgen_amd/synth.py;
    this one pins the loaders against genuine svm_save_model output (the trained mipgen_svr.model itself is absent upstream)."""
    R = po.refdrv()
    rng = np.random.default_rng(4242)
    dp = C.POINTER(C.c_double)
    g = genome

    def cand(t, clean):
        e = int(rng.integers(16, 31)); l = int(rng.integers(18, 31))
        ss = int(rng.integers(75, 215))
        p = int(rng.integers(400, len(g) - 700)); strand = int(rng.integers(0, 2))
        ext = bytearray(g[p - 1 - e:p - 1] if strand == 0 else g[p - 1 + ss:p - 1 + ss + e])
        ins = bytearray(g[p - 1:p - 1 + ss])
        lig = bytearray(g[p - 1 + ss:p - 1 + ss + l] if strand == 0 else g[p - 1 - l:p - 1])
        ec = int(rng.choice([1, 1, 1, 1, 2, 3, 7, 20])); lc = int(rng.choice([1, 1, 1, 1, 2, 5, 19]))
        if not clean:
            kind = t % 15
            if kind == 3: ext[int(rng.integers(0, e))] = ord("N")          # guard: all-zero vector
            if kind == 6: ins[int(rng.integers(0, ss))] = ord("N")
            if kind == 9: ec = 0                                           # log10(0) = -inf
            if kind == 12: lc = 101                                        # clamp to 2
        lrc = rng.uniform(0, 0.3, 44)
        x = np.empty(192)
        R.ref_parameters(strand, bytes(ext), bytes(lig), bytes(ins), ec, lc, MIDDLE, lrc.ctypes.data_as(dp), x.ctypes.data_as(dp))
        s = R.ref_logistic(strand, bytes(ext), bytes(lig), bytes(ins), ec, lc, MIDDLE)
        return x, s

    rows, ys = [], []
    while len(rows) < 360:
        x, s = cand(len(rows), True)
        if not np.all(np.isfinite(x)) or not np.any(x):
            continue
        rows.append(x); ys.append(1.4 + 2.2 * (s - 0.5) + 0.08 * rng.standard_normal())
    X = np.ascontiguousarray(np.array(rows)); y = np.ascontiguousarray(np.array(ys))
    path = os.path.join(HERE, "models", "svr_libsvm_trained.model")
    n_sv = R.ref_svm_train_save(X.shape[0], X.ctypes.data_as(dp), y.ctypes.data_as(dp), 2e-4, 8.0, 0.12, path.encode())
    assert n_sv > 20, n_sv
    m = R.ref_svm_load_model(path.encode())
    tx, tv = [], []
    for t in range(150):
        x, _ = cand(t, False)
        tx.append(x); tv.append(R.ref_predict_text(m, x.ctypes.data_as(dp), 192))
    np.savez_compressed(os.path.join(HERE, "libsvm_trained.npz"), params=np.array(tx), svr=np.array(tv), n_sv=np.array([R.ref_svm_nsv(m)]),
                        gamma=np.array([R.ref_svm_gamma(m)]), rho=np.array([R.ref_svm_rho(m)]))
    print(f"libsvm-trained model: {n_sv} support vectors of {X.shape[0]} training candidates, gamma {R.ref_svm_gamma(m)}, rho {R.ref_svm_rho(m)}; "
          f"{len(tv)} known answers in [{np.nanmin(tv):.3f}, {np.nanmax(tv):.3f}]")


DESIGNS = [
    # name, method, intervals, minC, maxC, sums, flank, tags, snps, trf, bwa_mode, model, keep_all
    dict(name="logistic_snp_trf", method="logistic", ivs=[("1", 5000, 5070, "a"), ("1", 9000, 9046, "b")], minC=120, maxC=125,
         sums=[40, 41], flank=5, tags="4,4", snps=True, trf=True, bwa="hashed", model=None, keep_all=True),
    dict(name="svr_small", method="svr", ivs=[("1", 5000, 5060, "a")], minC=130, maxC=140, sums=[44, 45], flank=0, tags="5,0",
         snps=False, trf=False, bwa="hashed", model="svr_syn_64.model", keep_all=True),
    dict(name="mixed_small", method="mixed", ivs=[("1", 5000, 5090, "a"), ("1", 5400, 5460, "b")], minC=125, maxC=135, sums=[42, 43],
         flank=0, tags="5,0", snps=True, trf=False, bwa="hashed", model="svr_syn_64.model", keep_all=True),
    dict(name="logistic_default_arms", method="logistic", ivs=[("1", 5000, 5120, "a"), ("1", 5300, 5420, "b"), ("1", 5440, 5500, "c")],
         minC=152, maxC=162, sums=[40, 41, 42, 43, 44, 45], flank=0, tags="5,0", snps=False, trf=False, bwa="unique", model=None,
         keep_all=False),
]


# Designs on the second golden genome (80 kb): longer regions, every selection-stage option, gaps, >= 10 regions on one chromosome.
# `extra` goes to the reference's command line verbatim; all_mips is kept as a sha256 of the normalised lines (tests/helpers.py).
DESIGNS2 = [
    dict(name="gaps_blocks", method="logistic", ivs=[("1", 6000, 8600, "long"), ("1", 12000, 12300, "b"), ("1", 13150, 13420, "c")], minC=152, maxC=162,
         sums=[40, 41, 42, 43, 44, 45], flank=0, tags="5,0", snps=False, trf=False, bwa="blocks", model=None, extra=[]),
    dict(name="double_tile_unaware", method="logistic", ivs=[("1", 20000, 20900, "a"), ("1", 22000, 22400, "b")], minC=152, maxC=162,
         sums=[40, 42, 44], flank=0, tags="5,0", snps=False, trf=False, bwa="hashed", model=None, extra=["-double_tile_strand_unaware", "on"]),
    dict(name="double_tile_separately", method="logistic", ivs=[("1", 24000, 24700, "a"), ("1", 25500, 25800, "b")], minC=152, maxC=162,
         sums=[41, 43, 45], flank=0, tags="5,0", snps=False, trf=False, bwa="hashed", model=None, extra=["-double_tile_strands_separately", "on"]),
    dict(name="seal_both", method="logistic", ivs=[("1", 28000, 28800, "a")], minC=140, maxC=150, sums=[40, 41, 42], flank=0, tags="5,0", snps=True,
         trf=False, bwa="hashed", model=None, extra=["-seal_both_strands", "on", "-starting_mip_overlap", "5"]),
    dict(name="half_seal_overlap", method="logistic", ivs=[("1", 30000, 30750, "a"), ("1", 31500, 31650, "b")], minC=140, maxC=150, sums=[40, 41, 42], flank=0,
         tags="5,0", snps=False, trf=True, bwa="hashed", model=None,
         extra=["-half_seal_both_strands", "on", "-max_mip_overlap", "40", "-starting_mip_overlap", "10", "-masked_arm_threshold", "0.2"]),
    dict(name="arm_lengths_unsorted", method="logistic", ivs=[("1", 34000, 34500, "a"), ("1", 35000, 35120, "b")], minC=150, maxC=160, sums=None,
         arm_lengths="20:22,16:24,25:20,18:27,22:20,30:18,17:25", flank=0, tags="5,0", snps=False, trf=False, bwa="hashed", model=None, extra=[]),
    dict(name="heuristic_off_copy_off", method="logistic", ivs=[("1", 38000, 38600, "a"), ("1", 39400, 39520, "b")], minC=152, maxC=162, sums=[40, 41, 42, 43, 44, 45],
         flank=0, tags="5,0", snps=False, trf=False, bwa="hashed", model=None, extra=["-logistic_heuristic", "off", "-check_copy_number", "off"]),
    dict(name="mixed_12_regions", method="mixed", ivs=[("1", 42000 + 700 * i, 42000 + 700 * i + 60 + 13 * i, f"m{i}") for i in range(12)], minC=125, maxC=135,
         sums=[42, 43], flank=0, tags="5,0", snps=True, trf=False, bwa="hashed", model="svr_syn_64.model", extra=[]),
    dict(name="svr_2kb", method="svr", ivs=[("1", 54000, 56000, "s")], minC=130, maxC=140, sums=[44, 45], flank=10, tags="4,4", snps=False, trf=False,
         bwa="hashed", model="svr_syn_64.model", extra=[]),
    dict(name="merge_flank_tags", method="logistic", ivs=[("1", 60000, 60100, "x"), ("1", 60160, 60300, "y"), ("1", 61000, 61090, "z")], minC=152, maxC=162,
         sums=[40, 41, 42, 43, 44, 45], flank=20, tags="0,8", snps=True, trf=True, bwa="hashed", model=None, extra=["-capture_increment", "2"]),
    dict(name="long_default", method="logistic", ivs=[("1", 64000, 67600, "big"), ("1", 69000, 69200, "small")], minC=120, maxC=160,
         sums=[40, 41, 42, 43, 44, 45], flank=0, tags="5,0", snps=True, trf=False, bwa="hashed", model=None, extra=["-capture_increment", "10"]),
    dict(name="double_tile_both", method="logistic", ivs=[("1", 72000, 72600, "a")], minC=152, maxC=162, sums=[42, 44], flank=0, tags="5,0", snps=False,
         trf=False, bwa="hashed", model=None, extra=["-double_tile_strand_unaware", "on", "-double_tile_strands_separately", "on"]),
    # the score limits, copy-number limits and minimum arm lengths as options (they change the arm-pair set, the replay's early exits, the
    # condense / collapse filters and the pick stage's thresholds)
    dict(name="limits_logistic", method="logistic", ivs=[("1", 74000, 74600, "a"), ("1", 75300, 75500, "b")], minC=150, maxC=160, sums=[40, 42, 44], flank=0,
         tags="5,0", snps=False, trf=False, bwa="hashed", model=None,
         extra=["-ext_min_length", "18", "-lig_min_length", "20", "-logistic_priority_score", "0.8", "-logistic_optimal_score", "0.93",
                "-max_arm_copy_product", "20", "-target_arm_copy", "5"]),
    # capture sizes above 1,100 bases: scan sizes beyond the 1,024 bases the accelerator's list kernels stage at a time (the reference holds std::strings)
    dict(name="long_capture_logistic", method="logistic", ivs=[("1", 10000, 10040, "L")], minC=1100, maxC=1110, sums=[40, 42, 44], flank=0, tags="5,0", snps=True,
         trf=False, bwa="hashed", model=None, extra=[]),
    dict(name="long_capture_svr", method="svr", ivs=[("1", 16000, 16030, "L")], minC=1100, maxC=1105, sums=[44, 45], flank=0, tags="5,0", snps=False,
         trf=False, bwa="hashed", model="svr_syn_64.model", extra=[]),
    # arm-length sums whose lists are EMPTY (30 < 16 + 18, 62 > 30 + 30: mipgen.cpp:245-258): the first scan position still uses the largest key (:421), and
    # the one list :434 never switches off is that of the smallest key - i.e. none; low optimal scores so that the switch-off happens
    dict(name="empty_sum_lists", method="logistic", ivs=[("1", 11000, 11300, "e"), ("1", 11900, 11960, "f")], minC=150, maxC=160, sums=[30, 41, 43, 62], flank=0,
         tags="5,0", snps=False, trf=False, bwa="hashed", model=None, extra=["-logistic_optimal_score", "0.9"]),
    dict(name="empty_sum_lists_svr", method="svr", ivs=[("1", 17000, 17090, "e")], minC=150, maxC=155, sums=[30, 44, 45, 61], flank=0,
         tags="5,0", snps=False, trf=False, bwa="hashed", model="svr_syn_64.model", extra=["-svr_optimal_score", "1.5"]),
    # no arm pair at all (39 < 18 + 22): the reference completes with header-only files and the coverage gaps
    dict(name="no_arm_pairs", method="logistic", ivs=[("1", 12000, 12150, "n"), ("1", 12400, 12420, "m")], minC=150, maxC=160, sums=[39], flank=3,
         tags="5,0", snps=True, trf=False, bwa="hashed", model=None, extra=["-ext_min_length", "18", "-lig_min_length", "22"]),
    # what the differential probe (tools/diff_probe.py) covers at random, pinned as fixed designs: both arm options at once with a pair given twice
    # (the lists merge, the pair is enumerated twice: mipgen.cpp:222-261); VCF records beyond biallelic SNVs; option values at the edges
    dict(name="both_arm_options", method="logistic", ivs=[("1", 13000, 13200, "a"), ("1", 13190, 13260, "b")], minC=150, maxC=160, sums=None,
         arm_lengths="20:22,25:20,20:22,18:27", flank=0, tags="5,0", snps=True, trf=False, bwa="hashed", model=None, extra=["-arm_length_sums", "41,44"]),
    dict(name="wild_vcf_mixed", method="mixed", ivs=[("1", 18000, 18120, "w")], minC=130, maxC=135, sums=[42, 43], flank=3, tags="4,4", snps=True, trf=False,
         bwa="hashed", model="svr_syn_64.model", extra=[], snp_hook="wild"),
    dict(name="edge_options", method="logistic", ivs=[("1", 19000, 19300, "x"), ("1", 21500, 21560, "y")], minC=150, maxC=160, sums=[40, 42, 44], flank=1000,
         tags="30,30", snps=True, trf=True, bwa="hashed", model=None,
         extra=["-target_arm_copy", "0", "-max_arm_copy_product", "100000000", "-masked_arm_threshold", "0", "-logistic_priority_score", "0.999",
                "-logistic_optimal_score", "0.5", "-max_mip_overlap", "200", "-starting_mip_overlap", "60"]),
    dict(name="edge_options_svr", method="svr", ivs=[("1", 23000, 23080, "x")], minC=150, maxC=160, sums=[44, 45], flank=0, tags="1,60", snps=False, trf=False,
         bwa="hashed", model="svr_syn_64.model", extra=["-svr_priority_score", "9", "-svr_optimal_score", "0", "-capture_increment", "50", "-seal_both_strands", "on"]),
    # the same kind of limits for the SVR, given through -file_of_parameters (mipgen.cpp:1445-1481)
    dict(name="limits_svr_parameter_file", method="svr", ivs=[("1", 77000, 77350, "s")], minC=130, maxC=140, sums=[44, 45], flank=0, tags="5,0", snps=False,
         trf=False, bwa="hashed", model="svr_syn_64.model", extra=[],
         params_file="# limits of the SVR design\n-svr_optimal_score 1.9\n-svr_priority_score 1.2\n-target_arm_copy 8\nnot an option line\n"),
    # scan sizes from 2 and -capture_increment 1: outside the tiled SVR kernel's limits (the list route of the accelerator), inside what the
    # reference accepts (mipgen.cpp:222-261, 427-444: any range / increment)
    # seventeen capture sizes = two runs of the dense SVR scorer: the front end's dynamic skip between the runs (mipgen.cpp:430; kernels_skip.hip) must
    # not change a byte - with this optimal score every position of the long region stops at the fourth capture size (some lists of it still constructed): the whole second run is skipped
    dict(name="svr_two_size_runs", method="svr", ivs=[("1", 30000, 30150, "t"), ("1", 31000, 31060, "u")], minC=120, maxC=200, sums=[42, 43, 44, 45], flank=0,
         tags="5,0", snps=False, trf=False, bwa="hashed", model="svr_syn_64.model", extra=["-svr_optimal_score", "2.8", "-svr_priority_score", "1.2"]),
    dict(name="svr_scan_size_2_increment_1", method="svr", ivs=[("1", 26000, 26045, "w")], minC=47, maxC=70, sums=[40, 41, 42, 43, 44, 45], flank=0,
         tags="5,0", snps=False, trf=False, bwa="hashed", model="svr_syn_short_48.model", extra=["-capture_increment", "1"]),
    # the model libsvm itself trained and wrote (gen_libsvm_trained_model): genuine svm_save_model output through svm_load_model's grammar, end to end
    dict(name="svr_libsvm_trained_model", method="svr", ivs=[("1", 46000, 46070, "lt"), ("1", 47200, 47290, "lu")], minC=135, maxC=150, sums=[43, 44, 45], flank=0,
         tags="5,0", snps=True, trf=False, bwa="hashed", model="svr_libsvm_trained.model", extra=[]),
]


# Designs over THREE chromosomes ("2", "10", "X": the reference sorts chromosome names as strings, mipgen.cpp:37-67 - "10" < "2" < "X"), BED files as
# users write them: unsorted, with and without the `chr` prefix (:1017), extra columns, space-separated fields, duplicate starts (the stable sort keeps
# their order; the later line's label wins the merge, :1019-1026), comment lines (:993).  The used-arm sets are per chromosome (:1925-1938).
BED_MULTI_A = """# three chromosomes, as a user's BED comes
chrX\t5200\t5290\txa\t0\t+
2\t6000\t6080
chr10\t7000\t7100\tten_a\t960
10\t7000\t7060\tten_dup
chr2\t5200\t5300\ttwo_a\t.\t-
X 9000 9070 xb
# a comment between the lines
chr2\t6050\t6120\ttwo_merge
10\t12000\t12090\tten_b\tx\ty\tz
chrX\t5330\t5400\txa2
"""
BED_MULTI_B = """chrX\t4000\t4060\tsx
10\t5000\t5055\tsten
chr2\t3000\t3062\tstwo
"""
BED_MULTI_C = """X\t20000\t20075\tmx1
chr2\t21000\t21060\tm2a
chr10\t22000\t22090\tm10a
2\t21100\t21160\tm2b
chrX\t20600\t20650\tmx2
10\t22400\t22470
"""
BED_MULTI_D = """chr10\t30000\t30400\td10
X\t31000\t31350\tdx
chr2\t32000\t32500\td2
2\t32900\t33000\td2b
"""
DESIGNS3 = [
    dict(name="multichr_logistic_snps", method="logistic", bed_text=BED_MULTI_A, minC=152, maxC=162, sums=[40, 41, 42, 43, 44, 45], flank=3, tags="4,4",
         snps=True, trf=False, bwa="hashed", model=None, extra=[]),
    dict(name="multichr_svr", method="svr", bed_text=BED_MULTI_B, minC=130, maxC=140, sums=[44, 45], flank=0, tags="5,0", snps=False, trf=False, bwa="hashed",
         model="svr_syn_64.model", extra=[]),
    dict(name="multichr_mixed", method="mixed", bed_text=BED_MULTI_C, minC=125, maxC=135, sums=[42, 43], flank=2, tags="4,4", snps=True, trf=True, bwa="hashed",
         model="svr_syn_64.model", extra=[]),
    dict(name="multichr_double_tile_separately", method="logistic", bed_text=BED_MULTI_D, minC=152, maxC=162, sums=[41, 43, 45], flank=0, tags="5,0", snps=False,
         trf=False, bwa="hashed", model=None, extra=["-double_tile_strands_separately", "on", "-seal_both_strands", "on"]),
]
MULTI_CHROMS = (("2", 40000, 302), ("10", 40000, 310), ("X", 40000, 388))       # (name, bases, seed)

# Designs on the HARD genome (mipgen_amd/synth.py: hard_genome, chromosome "4"): ambiguity codes, lower case, '-' bytes, homopolymers,
# microsatellites, 20 % / 70 % GC - in arms and inserts of emitted candidates, through every scorer; non-silent (all_mips compared record by record)
_Z = synth.HARD_ZONES
DESIGNS4 = [
    dict(name="hard_logistic", method="logistic", chrom="4",
         ivs=[("4", _Z["iupac"][0] + 600, _Z["iupac"][0] + 900, "iupac"), ("4", _Z["lower"][0] + 500, _Z["lower"][0] + 760, "lower"),
              ("4", _Z["lowcomplex"][0] + 700, _Z["lowcomplex"][0] + 1500, "lowcx"), ("4", _Z["gc20"][0] + 500, _Z["gc20"][0] + 700, "gc20"),
              ("4", _Z["gc70"][0] + 500, _Z["gc70"][0] + 700, "gc70"), ("4", _Z["dash"][0] + 300, _Z["dash"][0] + 900, "dash")],
         minC=152, maxC=162, sums=[40, 41, 42, 43, 44, 45], flank=0, tags="5,0", snps=True, trf=False, bwa="hashed", model=None, extra=[]),
    dict(name="hard_svr", method="svr", chrom="4",
         ivs=[("4", _Z["iupac"][0] + 2000, _Z["iupac"][0] + 2120, "iupac"), ("4", _Z["lower"][0] + 1800, _Z["lower"][0] + 1900, "lower"),
              ("4", _Z["lowcomplex"][0] + 2600, _Z["lowcomplex"][0] + 2900, "lowcx"), ("4", _Z["gc20"][0] + 900, _Z["gc20"][0] + 1000, "gc20"),
              ("4", _Z["gc70"][0] + 900, _Z["gc70"][0] + 1000, "gc70"), ("4", _Z["dash"][0] + 1100, _Z["dash"][0] + 1300, "dash")],
         minC=140, maxC=160, sums=[43, 44, 45], flank=0, tags="5,0", snps=False, trf=False, bwa="hashed", model="svr_syn_64.model", extra=[]),
    dict(name="hard_mixed", method="mixed", chrom="4",
         ivs=[("4", _Z["iupac"][0] + 3300, _Z["iupac"][0] + 3450, "iupac"), ("4", _Z["lower"][0] + 2700, _Z["lower"][0] + 2850, "lower"),
              ("4", _Z["lowcomplex"][0] + 4000, _Z["lowcomplex"][0] + 4400, "lowcx"), ("4", _Z["gc70"][0] + 1300, _Z["gc70"][0] + 1420, "gc70"),
              ("4", _Z["dash"][0] + 1500, _Z["dash"][0] + 1650, "dash")],
         minC=125, maxC=135, sums=[42, 43], flank=4, tags="4,4", snps=True, trf=True, bwa="hashed", model="svr_syn_64.model", extra=[]),
    # the low-complexity zone alone at the default capture range, every arm-length sum, silent mode like a whole-exome run
    dict(name="hard_lowcomplexity_svr_silent", method="svr", chrom="4",
         ivs=[("4", _Z["lowcomplex"][0] + 100, _Z["lowcomplex"][0] + 700, "lc1"), ("4", _Z["lowcomplex"][0] + 4700, _Z["lowcomplex"][0] + 5200, "lc2"),
              ("4", _Z["gc20"][0] + 1200, _Z["gc20"][0] + 1500, "gc20"), ("4", _Z["gc70"][0] + 1500, _Z["gc70"][0] + 1800, "gc70")],
         minC=152, maxC=162, sums=[40, 41, 42, 43, 44, 45], flank=0, tags="5,0", snps=False, trf=False, bwa="hashed", model="svr_syn_200.model",
         extra=["-silent_mode", "on"]),
    # SATURATED logistic scores (found by tools/diff_probe.py ... hard, designs 21076 and 21139): inside a (CCG)n run the exponent reaches 36.7-37.4, b^x
    # lies in [2^53, 2^54), and the reference's score - exactly 1.0, or one / two ulps below - turns on the LAST bit of its pow (1 + y is a tie rounded
    # to even); collapse / condense compare those doubles with a strict >.  The accelerator re-scores exactly those candidates in the reference's term
    # order with a correctly rounded power (mipgen_amd/csrc/pow_base_cr.h) before anything is replayed.
    dict(name="hard_saturated_logistic", method="logistic", chrom="4", ivs=[("4", 18743, 18893, "r0")], minC=160, maxC=180, sums=[40, 42, 45, 46], flank=3,
         tags="5,0", snps=True, trf=False, bwa="hashed", model=None, extra=["-seal_both_strands", "on"]),
    # EXACT TIES: regions inside the (CA)65, (GGC)45 and (TG)68 runs with every copy number 1 - candidates shifted by the repeat's period have identical arms and
    # inserts, so the reference scores them bit-identically and keeps the first (collapse / condense: strictly greater); the accelerator must tie them too
    dict(name="hard_ties_svr", method="svr", chrom="4", ivs=[("4", 17900, 17930, "ca"), ("4", 18710, 18750, "ggc")], minC=120, maxC=130, sums=[40, 42, 44], flank=0,
         tags="5,0", snps=False, trf=False, bwa="unique", model="svr_syn_200.model", extra=["-svr_optimal_score", "3.5"]),
    dict(name="hard_ties_mixed", method="mixed", chrom="4", ivs=[("4", 19530, 19600, "tg")], minC=120, maxC=130, sums=[40, 41, 42], flank=0, tags="5,0", snps=False,
         trf=False, bwa="unique", model="svr_syn_64.model", extra=[]),
    dict(name="hard_saturated_mixed", method="mixed", chrom="4", ivs=[("4", 18656, 18658, "r0")], minC=123, maxC=173, sums=[41, 43, 46], flank=3, tags="5,0",
         snps=False, trf=False, bwa="blocks", model="svr_syn_64.model",
         extra=["-double_tile_strand_unaware", "on", "-masked_arm_threshold", "0.1", "-target_arm_copy", "50", "-max_arm_copy_product", "400", "-lig_min_length", "18"]),
]


def parse_bed_text(text: str):
    ivs = []
    for line in text.split("\n"):
        line = line.strip()
        if len(line) <= 1 or line[0] == "#":
            continue
        f = line.split()
        c = f[0][3:] if f[0].startswith("chr") else f[0]
        ivs.append((c, int(f[1]), int(f[2]), f[3] if len(f) > 3 else ""))
    return ivs


def wild_snps(snps):
    """VCF records beyond biallelic SNVs, deterministically: every 5th record an insertion (ALT longer than REF), every 7th two ALT alleles, every 11th
    position listed a second time with another ALT, every 13th with a `chr` prefix on the chromosome column (parse_vcf keys by that column as it stands)."""
    out = []
    for i, s in enumerate(snps):
        if i % 5 == 1:
            s = synth.Snp(s.chrom, s.pos, s.ref, s.alt + "ACGT"[i % 4])
        elif i % 7 == 2:
            s = synth.Snp(s.chrom, s.pos, s.ref, s.alt + "," + "ACGT"[(i + 1) % 4])
        elif i % 13 == 3:
            s = synth.Snp("chr" + s.chrom, s.pos, s.ref, s.alt)
        if i % 11 == 4:
            out.append(synth.Snp(s.chrom, s.pos, s.ref, "ACGT"[(i + 2) % 4]))
        out.append(s)
    return out


def gen_design(genome, d: dict, genome_name: str = "genome_chr1.fa.gz", out_root: str = HERE) -> None:
    if d.get("snp_hook") == "wild":
        d = dict(d, snp_hook=wild_snps)
    out = os.path.join(out_root, "design_" + d["name"])
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(out)
    w = "/tmp/mipgen_golden_" + d["name"]
    shutil.rmtree(w, ignore_errors=True)
    os.makedirs(w + "/genome")
    chrom = d.get("chrom", "1")
    multi = isinstance(genome, dict)                        # several chromosomes: {name: bases}; the BED is given as text
    if multi:
        for c, g in genome.items():
            synth.write_fasta(w + f"/genome/chr{c}.fa", "chr" + c, g)
        with open(w + "/regions.bed", "w") as fh:
            fh.write(d["bed_text"])
        d = dict(d, ivs=parse_bed_text(d["bed_text"]))
        ivs = [synth.Interval(*iv) for iv in d["ivs"]]
    else:
        synth.write_fasta(w + f"/genome/chr{chrom}.fa", "chr" + chrom, genome)
        ivs = [synth.Interval(*iv) for iv in d["ivs"]]
        synth.write_bed(w + "/regions.bed", ivs)
    shutil.copy(w + "/regions.bed", out + "/regions.bed")
    extra = ["-feature_flank", str(d["flank"]), "-tag_sizes", d["tags"]]
    extra += ["-arm_lengths", d["arm_lengths"]] if d.get("arm_lengths") else ["-arm_length_sums", ",".join(map(str, d["sums"]))]
    extra += list(d.get("extra", []))
    if d.get("params_file"):
        with open(w + "/params.txt", "w") as fh:
            fh.write(d["params_file"])
        shutil.copy(w + "/params.txt", out + "/params.txt")
        extra += ["-file_of_parameters", w + "/params.txt"]
    snp_path = None
    if d["snps"] and multi:
        snps = []
        for k, (c, g) in enumerate(genome.items()):
            mine = [iv for iv in ivs if iv.chrom == c]
            if mine:
                snps += synth.random_snps(c, g, min(iv.bed_start for iv in mine) - 400, max(iv.bed_end for iv in mine) + 400, seed=13 + k, per_bp=1 / 50.0)
        snp_path = w + "/snps.vcf"
        synth.write_vcf(snp_path, d["snp_hook"](snps) if d.get("snp_hook") else snps)
        shutil.copy(snp_path, out + "/snps.vcf")
    elif d["snps"]:
        lo = min(iv.bed_start for iv in ivs) - 1000
        hi = max(iv.bed_end for iv in ivs) + 1000
        snps = synth.random_snps("1", genome, 4000, 10000, seed=13, per_bp=1 / 40.0) if genome_name == "genome_chr1.fa.gz" else \
            synth.random_snps(chrom, genome.upper(), lo, hi, seed=13, per_bp=1 / 60.0)
        snp_path = w + "/snps.vcf"
        synth.write_vcf(snp_path, d["snp_hook"](snps) if d.get("snp_hook") else snps)
        shutil.copy(snp_path, out + "/snps.vcf")
    model = os.path.join(HERE, "models", d["model"]) if d["model"] else None
    r = run_reference(w, w + "/genome", w + "/regions.bed", "out", d["minC"], d["maxC"], score_method=d["method"],
                      model_path=model, bwa_mode=d["bwa"], snp_file=snp_path, use_trf=d["trf"], extra=extra)
    assert r["returncode"] == 0, r["stderr"]
    meta = {k: d[k] for k in ("name", "method", "minC", "maxC", "sums", "flank", "tags", "snps", "trf", "bwa", "model")}
    meta["intervals"] = d["ivs"]
    meta["genome"] = genome_name
    if multi:
        meta["genomes"] = {c: f"genome3_chr{c}.fa.gz" for c in genome}
        meta["bed_as_given"] = True
    if genome_name != "genome_chr1.fa.gz":
        meta["extra"] = list(d.get("extra", []))
        meta["chrom"] = d.get("chrom", "1")
        meta["arm_lengths"] = d.get("arm_lengths")
        meta["params_file"] = bool(d.get("params_file"))
    meta["sha256"] = {}
    meta["lines"] = {}
    for key in ("all_mips", "collapsed_mips", "picked_mips", "snp_mips"):
        path = r[key]
        meta["sha256"][key] = sha256(path)
        with open(path, "rb") as fh:
            data = fh.read()
        meta["lines"][key] = data.count(b"\n")
        if key == "all_mips" and not d.get("keep_all", False):
            if genome_name != "genome_chr1.fa.gz":
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                from helpers import normalise_all_mips
                norm = normalise_all_mips(data)
                meta["sha256"]["all_mips_normalised"] = hashlib.sha256(norm).hexdigest()
                if norm.count(b"\n") != meta["lines"]["all_mips"]:          # (the reference's uninitialised flag byte was a newline somewhere)
                    meta["lines"]["all_mips_normalised"] = norm.count(b"\n")
            continue
        with gzip.GzipFile(out + f"/ref.{key}.txt.gz", "wb", mtime=0) as gz:
            gz.write(data)
    meta_gaps = []
    for gap in ("coverage_failed.bed", "double_tile_failed.bed", "minus_strand_failed.bed", "minus_strand_double_tile_failed.bed"):
        p = os.path.join(w, "out." + gap)
        if os.path.exists(p):
            shutil.copy(p, out + "/ref." + gap)
            meta_gaps.append(gap)
    if genome_name != "genome_chr1.fa.gz":
        meta["gap_files"] = meta_gaps
    with open(out + "/meta.json", "w") as fh:
        json.dump(meta, fh, indent=1)
    print("design", d["name"], meta["lines"])


def main() -> None:
    if not (po.have_refdrv() and os.path.exists(os.path.join(ROOT, "oracle", "_ref", "mipgen_ref"))):
        raise SystemExit("oracle/_ref is not built: run `make -C oracle all` where /root/reference exists")
    only = set(sys.argv[1:])                               # names of designs to (re)generate; none = everything
    genome = synth.random_genome(20000, 101, n_run_frac=0.003, n_run_len=6)
    if not only:
        with gzip.GzipFile(os.path.join(HERE, "genome_chr1.fa.gz"), "wb", mtime=0) as gz:
            gz.write(b">chr1\n")
            for i in range(0, len(genome), 60):
                gz.write(genome[i:i + 60] + b"\n")
        gen_models(genome)
        gen_candidates(genome)
    elif not os.path.exists(os.path.join(HERE, "models", "svr_syn_short_48.model")):
        gen_short_model(genome)
    if not only or "libsvm_trained" in only or not os.path.exists(os.path.join(HERE, "models", "svr_libsvm_trained.model")):
        gen_libsvm_trained_model(genome)
    for d in DESIGNS:
        if not only or d["name"] in only:
            gen_design(genome, d)
    genome2 = synth.random_genome(80000, 202, n_run_frac=0.002, n_run_len=8)
    with gzip.GzipFile(os.path.join(HERE, "genome2_chr1.fa.gz"), "wb", mtime=0) as gz:
        gz.write(b">chr1\n")
        for i in range(0, len(genome2), 60):
            gz.write(genome2[i:i + 60] + b"\n")
    for d in DESIGNS2:
        if not only or d["name"] in only:
            gen_design(genome2, d, "genome2_chr1.fa.gz")
    # BASELINE configs[0]: the practice62 stand-in for practice_genes.bed (62 exon-like regions on a 400 kb chromosome "7"), capture 162-162,
    # logistic scoring - the reference's own CPU-runnable case
    from mipgen_amd import workloads
    g7, ivs7 = workloads.practice62()
    with gzip.GzipFile(os.path.join(HERE, "genome_practice62_chr7.fa.gz"), "wb", mtime=0) as gz:
        gz.write(b">chr7\n")
        for i in range(0, len(g7), 60):
            gz.write(g7[i:i + 60] + b"\n")
    d1 = dict(name="practice62_config1", method="logistic", ivs=[(iv.chrom, iv.bed_start, iv.bed_end, iv.label) for iv in ivs7], minC=162, maxC=162,
              sums=[40, 41, 42, 43, 44, 45], flank=0, tags="5,0", snps=False, trf=False, bwa="unique", model=None, extra=[], chrom="7")
    if not only or d1["name"] in only:
        gen_design(g7, d1, "genome_practice62_chr7.fa.gz")
    # BASELINE configs[1]: the same 62 regions, capture 140-180, SVR scoring (a 64-SV synthetic model keeps the reference's run at minutes),
    # -silent_mode on: the 15.5 M all_mips records are not written, the collapsed / picked / snp files are what is compared
    d2 = dict(d1, name="practice62_config2_svr", method="svr", minC=140, maxC=180, model="svr_syn_64.model", extra=["-silent_mode", "on"])
    if not only or d2["name"] in only:
        gen_design(g7, d2, "genome_practice62_chr7.fa.gz")
    # three chromosomes
    multi = {c: synth.random_genome(n, seed, n_run_frac=0.002, n_run_len=7) for c, n, seed in MULTI_CHROMS}
    for c, g in multi.items():
        with gzip.GzipFile(os.path.join(HERE, f"genome3_chr{c}.fa.gz"), "wb", mtime=0) as gz:
            gz.write(f">chr{c}\n".encode())
            for i in range(0, len(g), 60):
                gz.write(g[i:i + 60] + b"\n")
    for d in DESIGNS3:
        if not only or d["name"] in only:
            gen_design(multi, d, "genome3")
    # the hard genome: chromosome "4", written with its lower case, ambiguity codes and '-' bytes as a FASTA file holds them
    genome4 = synth.hard_genome()
    with gzip.GzipFile(os.path.join(HERE, "genome4_chr4.fa.gz"), "wb", mtime=0) as gz:
        gz.write(b">chr4\n")
        for i in range(0, len(genome4), 60):
            gz.write(genome4[i:i + 60] + b"\n")
    if not only or "candidates_hard" in only:
        gen_candidates_hard(genome4)
    for d in DESIGNS4:
        if not only or d["name"] in only:
            gen_design(genome4, d, "genome4_chr4.fa.gz")

    # BASELINE configs[3] in small: the first 1,000 exons of the synthetic exome (chromosome "1" cut behind them), capture 150-170 (five sizes), 57 arm pairs,
    # -silent_mode on - the selection stage over a thousand neighbouring regions (2,000+ picks, the rand() stream, the used-arm sets from exon to exon), 1.4e8
    # dense candidates: 25 minutes of the reference per design
    cl, ex = workloads.exome_layout()
    ex = [iv for iv in ex if iv.chrom == "1"][:1000]
    genome5 = workloads.exome_chromosome("1", cl["1"])[:ex[-1].bed_end + 4400]
    if not only or any(n.startswith("exome1000") for n in only):
        with gzip.GzipFile(os.path.join(HERE, "genome5_chr1.fa.gz"), "wb", mtime=0) as gz:
            gz.write(b">chr1\n")
            for i in range(0, len(genome5), 60):
                gz.write(genome5[i:i + 60] + b"\n")
    d5 = dict(name="exome1000_logistic_silent", method="logistic", ivs=[(iv.chrom, iv.bed_start, iv.bed_end, iv.label) for iv in ex], minC=150, maxC=170,
              sums=[40, 41, 42, 43, 44, 45], flank=0, tags="5,0", snps=True, trf=True, bwa="hashed", model=None, extra=["-silent_mode", "on"], chrom="1")
    d6 = dict(d5, name="exome1000_mixed_silent", method="mixed", model="svr_syn_64.model", trf=False, bwa="blocks")
    for d in (d5, d6):
        if not only or d["name"] in only:
            gen_design(genome5, d, "genome5_chr1.fa.gz")


if __name__ == "__main__":
    main()

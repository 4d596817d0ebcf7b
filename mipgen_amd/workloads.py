"""Synthetic stand-ins for the BASELINE.json workloads, built with the product's own
host-side helpers.  Used by bench.py and the full-size property tests."""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import numpy as np

from . import capi, synth


def practice62(seed: int = 20140101, genome_len: int = 400_000, n_regions: int = 62) -> Tuple[bytes, List[synth.Interval]]:
    """`practice62`: one 400 kb chromosome "7", 62 exon-like intervals (EGFR/TERT/BRAF-like)."""
    genome = synth.random_genome(genome_len, seed)
    ivs = synth.practice62_intervals("7", seed=seed, n=n_regions)
    # keep every interval inside the chromosome with room for the +/- (max_capture + 1000) context
    ivs = [iv for iv in ivs if iv.bed_end + 1500 < genome_len]
    return genome, ivs


def svr_model_path(cache_dir: str, genome: bytes, n_sv: int, seed: int = 7) -> str:
    os.makedirs(cache_dir, exist_ok=True)
    path = os.path.join(cache_dir, f"svr_syn_{n_sv}_s{seed}.model")
    if not os.path.exists(path):
        tmp = path + f".tmp{os.getpid()}"
        synth.synthetic_svr_model(tmp, genome, n_sv, seed=seed)
        os.replace(tmp, path)
    return path


def build_regions(acc: Optional[capi.Accel], genome: bytes, ivs: List[synth.Interval], params: capi.Params,
                  bwa_mode: str = "unique", with_lrc: bool = True) -> List[capi.RegionData]:
    """Region records as the reference's -genome_dir input stage produces them; the 44 long-range k-mer frequencies
    come from the device kernel (mipgen_accel_long_range_content) when an accelerator handle is given."""
    out = []
    for iv in ivs:
        rd = capi.build_region(genome, iv.chrom, iv.bed_start, iv.bed_end, params, label=iv.label, bwa_mode=bwa_mode)
        if with_lrc and acc is not None:
            n = rd.c.seq_stop - rd.c.seq_start + 1
            s0 = rd.c.start_flanked - params.max_capture_size - 1 - 1000         # /root/reference/mipgen.cpp:1225
            lrc = acc.long_range_content(genome[s0:s0 + n + 2000], rd.c.seq_start, rd.c.seq_stop)
            for i in range(capi.N_LRC):
                rd.c.long_range_content[i] = lrc[i]
        out.append(rd)
    return out

"""ctypes binding of the C-ABI in include/mipgen_accel.h (libmipgen_accel.so, hand-written HIP for gfx950).

This is plumbing for tests and bench.py: Python is not the product's host language (the reference is C++,
and so is the host side under mipgen_amd/host/); everything here maps 1:1 onto the C entry points.
There is no CPU fallback: if the shared library is missing or no HIP device is usable, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libmipgen_accel.so")

MAX_ARM_PAIRS = 256
N_FEATURES = 192
N_LRC = 44
MAX_OLIGO = 64

SCORE_LOGISTIC, SCORE_SVR, SCORE_MIXED = 0, 1, 2

ABI_VERSION = 6          # include/mipgen_accel.h: MIPGEN_ACCEL_ABI_VERSION
FLAG_VALID, FLAG_GUARD, FLAG_MAPPING, FLAG_MASKING, FLAG_SNP, FLAG_HAS_SNP_MIP = 1, 2, 4, 8, 16, 32


class Params(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("score_method", C.c_int32),
        ("min_capture_size", C.c_int32), ("max_capture_size", C.c_int32), ("capture_increment", C.c_int32),
        ("max_mip_overlap", C.c_int32), ("n_arm_pairs", C.c_int32),
        ("arm_ext", C.c_int32 * MAX_ARM_PAIRS), ("arm_lig", C.c_int32 * MAX_ARM_PAIRS),
        ("check_copy_number", C.c_int32), ("logistic_heuristic", C.c_int32),
        ("masked_arm_threshold", C.c_double), ("upper_score_limit", C.c_double), ("lower_score_limit", C.c_double),
        ("max_arm_copy_product", C.c_int32), ("target_arm_copy", C.c_int32), ("arm_sum_key_max", C.c_int32), ("arm_sum_key_min", C.c_int32),
        ("reserved", C.c_int32 * 4),
    ]


class Region(C.Structure):
    _fields_ = [
        ("start_flanked", C.c_int32), ("stop_flanked", C.c_int32), ("seq_start", C.c_int32), ("seq_stop", C.c_int32),
        ("seq_len", C.c_int32), ("reserved0", C.c_int32),
        ("seq", C.c_char_p), ("masked_seq", C.c_char_p),
        ("copy", C.POINTER(C.POINTER(C.c_int32))), ("unmappable", C.POINTER(C.c_uint8)), ("snp_class", C.POINTER(C.c_uint8)),
        ("long_range_content", C.c_double * N_LRC),
    ]


class Grid(C.Structure):
    _fields_ = [("offset", C.c_int64), ("count", C.c_int64), ("first_pos", C.c_int32), ("n_pos", C.c_int32),
                ("first_size_index", C.c_int32), ("n_sizes", C.c_int32)]


class Candidate(C.Structure):
    _fields_ = [("region", C.c_int32), ("scan_start", C.c_int32), ("capture_size", C.c_int32),
                ("ext_len", C.c_int32), ("lig_len", C.c_int32), ("strand", C.c_int32)]


class CandidateInts(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "ext_a", "ext_c", "ext_g", "ext_t", "lig_a", "lig_c", "lig_g", "lig_t", "ins_a", "ins_c", "ins_g", "ins_t",
        "run_count", "junction", "ext_copy", "lig_copy", "masked_n", "snp_count", "flags", "scan_size")]


class Survivor(C.Structure):
    _fields_ = [("cand_index", C.c_int64), ("score", C.c_double), ("record", C.c_uint64)]


class RecordNames(C.Structure):
    _fields_ = [("chr", C.c_char_p), ("label", C.c_char_p), ("feature_start", C.c_int32), ("feature_stop", C.c_int32)]


class WindowViews(C.Structure):      # mipgen_window_views: device pointers of a result window
    _fields_ = [("emitted", C.c_void_p), ("survivors", C.c_void_p), ("collapsed", C.c_void_p), ("survivor_svr", C.c_void_p), ("text", C.c_void_p),
                ("n_emitted", C.c_int64), ("n_survivors", C.c_int64), ("n_collapsed", C.c_int64), ("n_text_bytes", C.c_int64), ("first_candidate", C.c_int64)]


SURVIVOR_DTYPE = np.dtype([("cand_index", "<i8"), ("score", "<f8"), ("record", "<u8")])
INTS_FIELDS = [f[0] for f in CandidateInts._fields_]


def make_params(min_capture: int, max_capture: int, score_method: int = SCORE_LOGISTIC, capture_increment: int = 5,
                max_mip_overlap: int = 30, arm_pairs: Optional[Sequence[Tuple[int, int]]] = None,
                check_copy_number: bool = True, logistic_heuristic: bool = True, masked_arm_threshold: float = 0.5,
                logistic_optimal: float = 0.98, logistic_priority: float = 0.9, svr_optimal: float = 2.2,
                svr_priority: float = 1.5, max_arm_copy_product: int = 75, target_arm_copy: int = 20,
                arm_sum_keys: Optional[Tuple[int, int]] = None) -> Params:
    """Defaults as mipgen::set_default_args / parse_arg_values (/root/reference/mipgen.cpp:164-188,209-216,243,264-265)."""
    from .synth import arm_pairs_from_sums
    p = Params()
    p.abi_version = ABI_VERSION
    p.score_method = score_method
    p.min_capture_size, p.max_capture_size = min_capture, max_capture
    p.capture_increment = capture_increment if capture_increment != 0 else 1
    p.max_mip_overlap = max_mip_overlap
    pairs = list(arm_pairs) if arm_pairs is not None else arm_pairs_from_sums()
    assert len(pairs) <= MAX_ARM_PAIRS
    p.n_arm_pairs = len(pairs)
    for i, (e, l) in enumerate(pairs):
        p.arm_ext[i], p.arm_lig[i] = e, l
    p.check_copy_number = int(check_copy_number)
    p.logistic_heuristic = int(logistic_heuristic)
    p.masked_arm_threshold = masked_arm_threshold
    svr = score_method == SCORE_SVR
    p.upper_score_limit = svr_optimal if svr else logistic_optimal
    p.lower_score_limit = svr_priority if svr else logistic_priority
    p.max_arm_copy_product = max_arm_copy_product
    p.target_arm_copy = target_arm_copy
    if arm_sum_keys is not None:                 # (largest, smallest) key of the reference's arm-sum map where such a key holds an empty list
        p.arm_sum_key_max, p.arm_sum_key_min = arm_sum_keys
    return p


def arm_pairs_of(p: Params) -> List[Tuple[int, int]]:
    return [(p.arm_ext[i], p.arm_lig[i]) for i in range(p.n_arm_pairs)]


COPY_RESIDENT = "resident"


class BigCopy(C.Structure):
    _fields_ = [("region", C.c_int32), ("length", C.c_int32), ("start", C.c_int32), ("copies", C.c_int32)]


class RegionData:
    """Owns the host arrays a mipgen_region points into (keeps them alive for ctypes)."""

    def __init__(self, start_flanked: int, stop_flanked: int, seq_start: int, seq: bytes,
                 masked: Optional[bytes] = None, copy=None,
                 unmappable: Optional[np.ndarray] = None, snp_class: Optional[np.ndarray] = None,
                 lrc: Optional[Sequence[float]] = None, chrom: str = "1", label: str = "x",
                 start: Optional[int] = None, stop: Optional[int] = None):
        self.chrom, self.label = chrom, label
        self.start = start if start is not None else start_flanked      # unflanked, for print_details
        self.stop = stop if stop is not None else stop_flanked
        self.seq = bytes(seq)
        self.masked = bytes(masked) if masked is not None else None
        resident = isinstance(copy, str)                                  # COPY_RESIDENT: the tables the handle counted itself
        assert not resident or copy == COPY_RESIDENT
        self.copy = {} if resident else {k: np.ascontiguousarray(v, dtype=np.int32) for k, v in (copy or {}).items()}
        self.unmappable = np.ascontiguousarray(unmappable, dtype=np.uint8) if unmappable is not None else None
        self.snp_class = np.ascontiguousarray(snp_class, dtype=np.uint8) if snp_class is not None else None
        self.alleles: Optional[bytes] = None                              # oracle-only allele table
        self.c = Region()
        r = self.c
        r.start_flanked, r.stop_flanked = start_flanked, stop_flanked
        r.seq_start = seq_start
        r.seq_len = len(self.seq)
        r.seq_stop = seq_start + len(self.seq) - 1
        r.seq = self.seq
        r.masked_seq = self.masked if self.masked is not None else None
        if resident:
            r.copy = C.cast(C.c_void_p(1), C.POINTER(C.POINTER(C.c_int32)))  # MIPGEN_COPY_RESIDENT
        elif copy is not None:
            self._copy_tab = (C.POINTER(C.c_int32) * (MAX_OLIGO + 1))()
            for k, v in self.copy.items():
                assert v.shape == (r.seq_len,)
                self._copy_tab[k] = v.ctypes.data_as(C.POINTER(C.c_int32))
            r.copy = C.cast(self._copy_tab, C.POINTER(C.POINTER(C.c_int32)))
        if self.unmappable is not None:
            r.unmappable = self.unmappable.ctypes.data_as(C.POINTER(C.c_uint8))
        if self.snp_class is not None:
            assert self.snp_class.shape == (r.seq_len,)
            r.snp_class = self.snp_class.ctypes.data_as(C.POINTER(C.c_uint8))
        if lrc is not None:
            for i in range(N_LRC):
                r.long_range_content[i] = float(lrc[i])


def region_array(regions: Sequence[RegionData]):
    arr = (Region * len(regions))()
    for i, r in enumerate(regions):
        arr[i] = r.c
    return arr


def n_sizes_all(p: Params) -> int:
    if p.max_capture_size < p.min_capture_size:
        return 0
    return (p.max_capture_size - p.min_capture_size) // p.capture_increment + 1


def build_region(genome: bytes, chrom: str, bed_start: int, bed_end: int, params: Params, flank: int = 0,
                 label: str = "x", bwa_mode: str = "unique", snp_tab: Optional[Dict[int, str]] = None,
                 mask_record: Optional[int] = None, lrc: Optional[Sequence[float]] = None,
                 pad: int = 15) -> RegionData:
    """Host-side construction of one region exactly as the reference's -genome_dir input path lays it out
    (/root/reference/mipgen.cpp:1180-1229: sequence = [max(1, start_fl - maxC), min(len, stop_fl + maxC + 15)])
    with the lookup tables the external-tool stand-ins imply (see synth.shim_*)."""
    from . import synth
    start = bed_start + 1
    stop = bed_end
    sf, ef = start - flank, stop + flank
    maxC = params.max_capture_size
    cs = max(1, sf - maxC)
    ce = min(len(genome), ef + maxC + pad)
    seq = genome[cs - 1:ce].upper()
    n = len(seq)
    pairs = arm_pairs_of(params)
    sizes = sorted({e for e, _ in pairs} | {l for _, l in pairs})
    copy = None
    unmappable = None
    if bwa_mode != "unique":
        copy = {}
        starts = np.arange(cs, cs + n, dtype=np.int64)
        for k in sizes:
            c = synth.shim_copy(starts, k, bwa_mode)
            # the reference only writes oligos with relative start < len - size (mipgen.cpp:829): later keys are absent -> 0
            c[max(0, n - k):] = 0
            copy[k] = c
        K = n_sizes_all(params)
        unmappable = np.zeros((K, n), dtype=np.uint8)
        for k in range(K):
            size = maxC - k * params.capture_increment
            # capture windows are written for start in [sf - size, ef) with start > 0 and start+size-1 <= ce (mipgen.cpp:812-823)
            pos = np.arange(cs, cs + n, dtype=np.int64)
            ok = (pos >= sf - size) & (pos < ef) & (pos > 0) & (pos + size - 1 <= ce)
            unmappable[k] = (synth.shim_unmappable(pos, size, bwa_mode) & ok).astype(np.uint8)
    else:
        # unique mode: table built by the reference still lacks the tail oligos -> model exactly
        copy = {}
        for k in sizes:
            c = np.ones(n, dtype=np.int32)
            c[max(0, n - k):] = 0
            copy[k] = c
    masked = synth.shim_mask(seq, mask_record) if mask_record is not None else None
    snp_class = None
    alleles = None
    if snp_tab:
        snp_class = np.zeros(n, dtype=np.uint8)
        al = bytearray(2 * n)
        comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
        for pos, a in snp_tab.items():
            i = pos - cs
            if 0 <= i < n:
                g = chr(seq[i])
                okc = (len(a) == 2 and a[0] not in "N-" and a[1] not in "N-" and (g == a[0] or (a[0] in comp and g == comp[a[0]])))
                snp_class[i] = 1 if okc else 2
                if len(a) == 2:
                    al[2 * i], al[2 * i + 1] = ord(a[0]), ord(a[1])
                else:
                    al[2 * i], al[2 * i + 1] = ord("*"), ord("*")
        alleles = bytes(al)
    rd = RegionData(sf, ef, cs, seq, masked=masked, copy=copy, unmappable=unmappable, snp_class=snp_class,
                    lrc=lrc, chrom=chrom, label=label, start=start, stop=stop)
    rd.alleles = alleles
    return rd


# ----------------------------------------------------------------------------------------------------
# the library
# ----------------------------------------------------------------------------------------------------

class AccelError(RuntimeError):
    pass


_lib = None


def load_library(path: Optional[str] = None):
    """dlopen libmipgen_accel.so and declare every prototype of include/mipgen_accel.h."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise AccelError(f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                         "(there is no CPU fallback)")
    lib = C.CDLL(p)
    vp = C.c_void_p
    lib.mipgen_accel_abi_version.restype = C.c_int
    lib.mipgen_accel_last_error.restype = C.c_char_p
    lib.mipgen_accel_device_count.restype = C.c_int
    lib.mipgen_accel_create.argtypes = [C.POINTER(Params), C.c_int, vp, C.POINTER(vp)]
    lib.mipgen_accel_destroy.argtypes = [vp]
    lib.mipgen_accel_destroy.restype = None
    lib.mipgen_accel_load_model_file.argtypes = [vp, C.c_char_p]
    lib.mipgen_accel_set_model.argtypes = [vp, C.c_int32, C.c_double, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.mipgen_accel_model_info.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.mipgen_accel_upload_regions.argtypes = [vp, C.POINTER(Region), C.c_int32, C.POINTER(Grid)]
    lib.mipgen_accel_batch_candidates.argtypes = [vp]
    lib.mipgen_accel_batch_candidates.restype = C.c_int64
    lib.mipgen_accel_score_resident.argtypes = [vp, C.c_int32]
    lib.mipgen_accel_result_device_ptrs.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    lib.mipgen_accel_download_results.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int64, C.c_int64]
    lib.mipgen_accel_score_regions.argtypes = [vp, C.POINTER(Region), C.c_int32, C.c_int32, C.POINTER(Grid),
                                               C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int64]
    lib.mipgen_accel_score_candidates.argtypes = [vp, C.POINTER(Candidate), C.c_int32, C.c_int32, C.POINTER(C.c_double),
                                                  C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.POINTER(CandidateInts)]
    lib.mipgen_accel_long_range_content.argtypes = [vp, C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_double)]
    lib.mipgen_accel_replay_condense.argtypes = [vp]
    lib.mipgen_accel_download_replay.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(Survivor), C.c_int64,
                                                 C.POINTER(C.c_uint8), C.c_int64]
    lib.mipgen_accel_last_kernel_ms.argtypes = [vp, C.c_int32]
    lib.mipgen_accel_last_kernel_ms.restype = C.c_double
    lib.mipgen_accel_set_timing.argtypes = [vp, C.c_int32]
    lib.mipgen_accel_set_window_candidates.argtypes = [vp, C.c_int64]
    lib.mipgen_accel_set_window_breaks.argtypes = [vp, C.POINTER(C.c_int32), C.c_int32]
    lib.mipgen_accel_window_count.argtypes = [vp]
    lib.mipgen_accel_window_count.restype = C.c_int32
    i32p, i64p = C.POINTER(C.c_int32), C.POINTER(C.c_int64)
    lib.mipgen_accel_window_info.argtypes = [vp, C.c_int32, i32p, i32p, i64p, i64p, i64p, i64p]
    lib.mipgen_accel_score_window.argtypes = [vp, C.c_int32, C.c_int32]
    lib.mipgen_accel_score_condense_all.argtypes = [vp, C.c_int32]
    lib.mipgen_accel_score_condense_window.argtypes = [vp, C.c_int32, C.c_int32]
    lib.mipgen_accel_download_survivors.argtypes = [vp, i64p, C.POINTER(Survivor), C.c_int64]
    lib.mipgen_accel_survivors_device_ptr.argtypes = [vp, C.POINTER(vp), i64p]
    lib.mipgen_accel_set_sv_split.argtypes = [vp, C.c_int32]
    lib.mipgen_accel_set_print_exact.argtypes = [vp, C.c_int32]
    lib.mipgen_accel_set_logistic_subruns.argtypes = [vp, C.c_int32]
    lib.mipgen_accel_collapse.argtypes = [vp]
    lib.mipgen_accel_region_bases.argtypes = [vp, C.c_int32, i64p, i32p]
    lib.mipgen_accel_download_collapsed.argtypes = [vp, C.c_int32, i32p, C.c_int64]
    lib.mipgen_accel_count_oligo_copies.argtypes = [vp, C.c_int32, C.POINTER(C.c_char_p), i64p, C.c_int32, C.POINTER(C.c_char_p), i32p, C.c_int32, i32p,
                                                    C.POINTER(C.POINTER(C.c_int32))]
    lib.mipgen_accel_count_oligo_copies_resident.argtypes = [vp, C.c_int32, C.POINTER(C.c_char_p), i64p, C.c_int32, C.POINTER(C.c_char_p), i32p, i64p,
                                                             C.POINTER(C.POINTER(BigCopy))]
    lib.mipgen_accel_window_uniqueness.argtypes = [vp, C.c_int32, C.POINTER(C.c_char_p), i64p, C.c_int32, C.POINTER(C.c_char_p), i32p, C.c_int32, i32p, C.c_int32,
                                                   C.POINTER(C.POINTER(C.c_uint8))]
    lib.mipgen_accel_window_uniqueness_begin.argtypes = [vp, C.c_int32, C.POINTER(C.c_char_p), i64p, C.c_int32, C.POINTER(C.c_char_p), i32p, i32p, C.c_int32, i32p,
                                                         C.c_int32, C.POINTER(C.c_uint8)]
    lib.mipgen_accel_window_flags_region.argtypes = [vp, C.c_int32, C.POINTER(C.c_uint8)]
    lib.mipgen_accel_window_uniqueness_end.argtypes = [vp]
    lib.mipgen_accel_set_dynamic_skip.argtypes = [vp, C.c_int32]
    lib.mipgen_accel_skipped_candidates.argtypes = [vp, i64p]
    lib.mipgen_accel_skip_state.argtypes = [vp, C.POINTER(C.c_uint8), C.POINTER(C.c_double), C.c_int64]
    lib.mipgen_accel_format_all_mips.argtypes = [vp, C.POINTER(RecordNames), C.c_char_p, C.c_int64, i64p, i64p]
    lib.mipgen_accel_download_text.argtypes = [vp, C.c_char_p, C.c_int64]
    lib.mipgen_accel_long_range_content_batch.argtypes = [vp, C.c_int32, C.POINTER(C.c_char_p), i32p, i32p, i32p, C.POINTER(C.c_double)]
    lib.mipgen_accel_rescore_survivors.argtypes = [vp]
    lib.mipgen_accel_download_survivor_scores.argtypes = [vp, C.c_int32, C.POINTER(C.c_double), C.c_int64]
    lib.mipgen_accel_window_views.argtypes = [vp, C.c_int32, C.POINTER(WindowViews)]
    lib.mipgen_accel_synchronize.argtypes = [vp]
    for name in ("rescore_survivors", "download_survivor_scores", "window_views", "synchronize"):
        getattr(lib, "mipgen_accel_" + name).restype = C.c_int
    for name in ("create", "load_model_file", "set_model", "model_info", "upload_regions", "score_resident",
                 "result_device_ptrs", "download_results", "score_regions", "score_candidates",
                 "long_range_content", "replay_condense", "download_replay", "set_timing", "set_window_candidates",
                 "window_info", "score_window", "score_condense_all", "score_condense_window", "download_survivors", "survivors_device_ptr",
                 "set_sv_split", "set_print_exact", "set_logistic_subruns", "long_range_content_batch", "collapse", "region_bases", "download_collapsed", "count_oligo_copies", "count_oligo_copies_resident", "window_uniqueness", "window_uniqueness_begin", "window_flags_region", "window_uniqueness_end",
                 "format_all_mips", "download_text", "set_dynamic_skip", "skipped_candidates", "skip_state"):
        getattr(lib, "mipgen_accel_" + name).restype = C.c_int
    if path is None:
        _lib = lib
    return lib


EXPORTED_SYMBOLS = [
    "mipgen_accel_abi_version", "mipgen_accel_last_error", "mipgen_accel_device_count", "mipgen_accel_create",
    "mipgen_accel_destroy", "mipgen_accel_load_model_file", "mipgen_accel_set_model", "mipgen_accel_model_info",
    "mipgen_accel_upload_regions", "mipgen_accel_batch_candidates", "mipgen_accel_score_resident",
    "mipgen_accel_result_device_ptrs", "mipgen_accel_download_results", "mipgen_accel_score_regions",
    "mipgen_accel_score_candidates", "mipgen_accel_long_range_content", "mipgen_accel_replay_condense",
    "mipgen_accel_download_replay", "mipgen_accel_last_kernel_ms", "mipgen_accel_set_timing",
    "mipgen_accel_set_window_candidates", "mipgen_accel_set_window_breaks", "mipgen_accel_window_count", "mipgen_accel_window_info", "mipgen_accel_score_window",
    "mipgen_accel_score_condense_all", "mipgen_accel_score_condense_window", "mipgen_accel_download_survivors", "mipgen_accel_survivors_device_ptr",
    "mipgen_accel_set_sv_split", "mipgen_accel_long_range_content_batch", "mipgen_accel_collapse", "mipgen_accel_region_bases",
    "mipgen_accel_download_collapsed", "mipgen_accel_count_oligo_copies", "mipgen_accel_format_all_mips", "mipgen_accel_download_text",
    "mipgen_accel_count_oligo_copies_resident", "mipgen_accel_window_uniqueness", "mipgen_accel_window_uniqueness_begin", "mipgen_accel_window_flags_region",
    "mipgen_accel_window_uniqueness_end", "mipgen_accel_set_dynamic_skip", "mipgen_accel_skipped_candidates", "mipgen_accel_skip_state", "mipgen_accel_set_print_exact", "mipgen_accel_set_logistic_subruns",
    "mipgen_accel_rescore_survivors", "mipgen_accel_download_survivor_scores", "mipgen_accel_window_views", "mipgen_accel_synchronize",
]


class Accel:
    """Thin RAII wrapper around a mipgen_accel handle."""

    def __init__(self, params: Params, device: int = 0, stream: int = 0):
        self.lib = load_library()
        self.params = params
        self.h = C.c_void_p()
        self._check(self.lib.mipgen_accel_create(C.byref(params), device, C.c_void_p(stream), C.byref(self.h)))
        self.grids: List[Grid] = []
        self._regions: Sequence[RegionData] = []

    def _check(self, rc: int) -> None:
        if rc != 0:
            msg = self.lib.mipgen_accel_last_error()
            raise AccelError(f"mipgen_accel error {rc}: {msg.decode() if msg else ''}")

    def close(self) -> None:
        if self.h:
            self.lib.mipgen_accel_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # model
    def load_model_file(self, path: str) -> None:
        self._check(self.lib.mipgen_accel_load_model_file(self.h, path.encode()))

    def set_model(self, gamma: float, rho: float, coef: np.ndarray, sv: np.ndarray) -> None:
        coef = np.ascontiguousarray(coef, dtype=np.float64)
        sv = np.ascontiguousarray(sv, dtype=np.float64)
        assert sv.shape == (coef.shape[0], N_FEATURES)
        self._check(self.lib.mipgen_accel_set_model(self.h, coef.shape[0], gamma, rho,
                                                    coef.ctypes.data_as(C.POINTER(C.c_double)),
                                                    sv.ctypes.data_as(C.POINTER(C.c_double))))

    def model_info(self) -> Tuple[int, float, float]:
        n, g, r = C.c_int32(), C.c_double(), C.c_double()
        self._check(self.lib.mipgen_accel_model_info(self.h, C.byref(n), C.byref(g), C.byref(r)))
        return n.value, g.value, r.value

    # regions
    def upload(self, regions: Sequence[RegionData]) -> List[Grid]:
        arr = region_array(regions)
        grids = (Grid * len(regions))()
        self._check(self.lib.mipgen_accel_upload_regions(self.h, arr, len(regions), grids))
        self.grids = list(grids)
        self._regions = regions
        return self.grids

    def upload_array(self, arr, n: int) -> List[Grid]:
        """upload() for a ready-made ctypes array of Region (hostapi.Design.regions)."""
        grids = (Grid * max(n, 1))()
        self._check(self.lib.mipgen_accel_upload_regions(self.h, arr, n, grids))
        self.grids = list(grids)[:n]
        self._regions = arr
        return self.grids

    def batch_candidates(self) -> int:
        return int(self.lib.mipgen_accel_batch_candidates(self.h))

    def score_resident(self, method: int) -> None:
        self._check(self.lib.mipgen_accel_score_resident(self.h, method))

    # result windows (batches whose dense results exceed the result arrays)
    def set_window_candidates(self, max_candidates: int) -> None:
        self._check(self.lib.mipgen_accel_set_window_candidates(self.h, max_candidates))

    def set_window_breaks(self, first_regions: Sequence[int]) -> None:
        """Batch indices at which a result window of the following uploads must start whatever the candidate bound says (ABI 5)."""
        arr = (C.c_int32 * max(len(first_regions), 1))(*first_regions)
        self._check(self.lib.mipgen_accel_set_window_breaks(self.h, arr, len(first_regions)))

    def window_count(self) -> int:
        return int(self.lib.mipgen_accel_window_count(self.h))

    def window_info(self, w: int) -> Dict[str, int]:
        r0, nr = C.c_int32(), C.c_int32()
        c0, nc, p0, np_ = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        self._check(self.lib.mipgen_accel_window_info(self.h, w, C.byref(r0), C.byref(nr), C.byref(c0), C.byref(nc), C.byref(p0), C.byref(np_)))
        return {"first_region": r0.value, "n_regions": nr.value, "first_candidate": c0.value, "n_candidates": nc.value,
                "first_position": p0.value, "n_positions": np_.value}

    def score_window(self, w: int, method: int) -> None:
        self._check(self.lib.mipgen_accel_score_window(self.h, w, method))

    def score_condense_all(self, method: int) -> None:
        self._check(self.lib.mipgen_accel_score_condense_all(self.h, method))

    def score_condense_window(self, w: int, method: int) -> None:
        """One window scored, replayed and condensed for a caller that never reads its dense results (ABI 6): the print-exact re-score tests the survivors only."""
        self._check(self.lib.mipgen_accel_score_condense_window(self.h, w, method))

    def download_survivors(self):
        nreg = len(self.grids)
        emitted = np.zeros(nreg, dtype=np.int64)
        npos = sum(g.n_pos for g in self.grids)
        surv = np.zeros(2 * npos, dtype=SURVIVOR_DTYPE)
        self._check(self.lib.mipgen_accel_download_survivors(self.h, emitted.ctypes.data_as(C.POINTER(C.c_int64)),
                                                             surv.ctypes.data_as(C.POINTER(Survivor)), 2 * npos))
        return emitted, surv

    def count_oligo_copies(self, chroms: Sequence[bytes], region_seqs: Sequence[bytes], lengths: Sequence[int]) -> List[Dict[int, np.ndarray]]:
        """Exact occurrence counts (both strands) of every oligo of the region strings in the genome: per region {length: int32[len(seq)]},
        the layout of mipgen_region.copy."""
        nc, nr, nl = len(chroms), len(region_seqs), len(lengths)
        ca = (C.c_char_p * max(nc, 1))(*chroms)
        cl = np.array([len(c) for c in chroms], dtype=np.int64)
        ra = (C.c_char_p * max(nr, 1))(*region_seqs)
        rl = np.array([len(r) for r in region_seqs], dtype=np.int32)
        la = np.ascontiguousarray(lengths, dtype=np.int32)
        outs = [np.zeros((nl, len(r)), dtype=np.int32) for r in region_seqs]
        op = (C.POINTER(C.c_int32) * max(nr, 1))(*[o.ctypes.data_as(C.POINTER(C.c_int32)) for o in outs])
        self._check(self.lib.mipgen_accel_count_oligo_copies(self.h, nc, ca, cl.ctypes.data_as(C.POINTER(C.c_int64)), nr, ra,
                                                             rl.ctypes.data_as(C.POINTER(C.c_int32)), nl, la.ctypes.data_as(C.POINTER(C.c_int32)), op))
        return [{int(k): o[i] for i, k in enumerate(lengths)} for o in outs]

    def window_uniqueness(self, chroms: Sequence[bytes], region_seqs: Sequence[bytes], sizes: Sequence[int], seed_len: int = 30) -> List[np.ndarray]:
        """Per region uint8 [len(sizes)][len(seq)]: 1 = the capture window of that size starting there is not unique within one substitution
        (both strands, whole genome) or holds a non-ACGT byte (mipgen.cpp:841-868 through bwa)."""
        nc, nr, ns = len(chroms), len(region_seqs), len(sizes)
        ca = (C.c_char_p * max(nc, 1))(*chroms)
        cl = np.array([len(c) for c in chroms], dtype=np.int64)
        ra = (C.c_char_p * max(nr, 1))(*region_seqs)
        rl = np.array([len(r) for r in region_seqs], dtype=np.int32)
        sz = np.array(list(sizes), dtype=np.int32)
        outs = [np.zeros((ns, len(r)), dtype=np.uint8) for r in region_seqs]
        op = (C.POINTER(C.c_uint8) * max(nr, 1))(*[o.ctypes.data_as(C.POINTER(C.c_uint8)) for o in outs])
        self._check(self.lib.mipgen_accel_window_uniqueness(self.h, nc, ca, cl.ctypes.data_as(C.POINTER(C.c_int64)), nr, ra,
                                                            rl.ctypes.data_as(C.POINTER(C.c_int32)), ns, sz.ctypes.data_as(C.POINTER(C.c_int32)), seed_len, op))
        return outs

    def set_dynamic_skip(self, on: bool) -> None:
        """mipgen.cpp:430 applied between the capture-size runs of the dense SVR scorer (the dense scores of skipped tiles read NaN)."""
        self._check(self.lib.mipgen_accel_set_dynamic_skip(self.h, int(bool(on))))

    def skip_state(self, n_pos: int) -> Tuple[np.ndarray, np.ndarray]:
        st = np.zeros(n_pos, dtype=np.uint8); pb = np.zeros(n_pos, dtype=np.float64)
        self._check(self.lib.mipgen_accel_skip_state(self.h, st.ctypes.data_as(C.POINTER(C.c_uint8)), pb.ctypes.data_as(C.POINTER(C.c_double)), n_pos))
        return st, pb

    def skipped_candidates(self) -> int:
        n = C.c_int64(0)
        self._check(self.lib.mipgen_accel_skipped_candidates(self.h, C.byref(n)))
        return int(n.value)

    def window_uniqueness_bounded(self, chroms: Sequence[bytes], region_seqs: Sequence[bytes], bounds: Sequence[Tuple[int, int, int, int]],
                                  sizes: Sequence[int], seed_len: int = 30) -> Tuple[np.ndarray, List[Optional[np.ndarray]]]:
        """mipgen_accel_window_uniqueness_begin / _flags_region / _end: the same flags restricted on the device to the window starts the reference
        looks up (bounds = (start_flanked, stop_flanked, seq_start, seq_stop) per region); returns the per-region any flags and the image of
        every region that has a flagged start (None for the others)."""
        nc, nr, ns = len(chroms), len(region_seqs), len(sizes)
        ca = (C.c_char_p * max(nc, 1))(*chroms)
        cl = np.array([len(c) for c in chroms], dtype=np.int64)
        ra = (C.c_char_p * max(nr, 1))(*region_seqs)
        rl = np.array([len(r) for r in region_seqs], dtype=np.int32)
        sz = np.array(list(sizes), dtype=np.int32)
        bd = np.ascontiguousarray(np.array(bounds, dtype=np.int32).reshape(nr, 4))
        any_ = np.zeros(max(nr, 1), dtype=np.uint8)
        i32 = C.POINTER(C.c_int32)
        self._check(self.lib.mipgen_accel_window_uniqueness_begin(self.h, nc, ca, cl.ctypes.data_as(C.POINTER(C.c_int64)), nr, ra, rl.ctypes.data_as(i32),
                                                                  bd.ctypes.data_as(i32), ns, sz.ctypes.data_as(i32), seed_len,
                                                                  any_.ctypes.data_as(C.POINTER(C.c_uint8))))
        outs: List[Optional[np.ndarray]] = []
        for r in range(nr):
            if not any_[r]:
                outs.append(None)
                continue
            o = np.zeros((ns, len(region_seqs[r])), dtype=np.uint8)
            self._check(self.lib.mipgen_accel_window_flags_region(self.h, r, o.ctypes.data_as(C.POINTER(C.c_uint8))))
            outs.append(o)
        self._check(self.lib.mipgen_accel_window_uniqueness_end(self.h))
        return any_[:nr], outs

    def count_oligo_copies_resident(self, chroms: Sequence[bytes], region_seqs: Sequence[bytes]) -> List[Tuple[int, int, int, int]]:
        """The same counts for the handle's own oligo lengths, left on the device for an upload of the same regions with copy=COPY_RESIDENT;
        returns the (region, length, start, copies) entries with 65535 copies or more."""
        nc, nr = len(chroms), len(region_seqs)
        ca = (C.c_char_p * max(nc, 1))(*chroms)
        cl = np.array([len(c) for c in chroms], dtype=np.int64)
        ra = (C.c_char_p * max(nr, 1))(*region_seqs)
        rl = np.array([len(r) for r in region_seqs], dtype=np.int32)
        nb = C.c_int64(0)
        big = C.POINTER(BigCopy)()
        self._check(self.lib.mipgen_accel_count_oligo_copies_resident(self.h, nc, ca, cl.ctypes.data_as(C.POINTER(C.c_int64)), nr, ra,
                                                                      rl.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(nb), C.byref(big)))
        return [(big[i].region, big[i].length, big[i].start, big[i].copies) for i in range(nb.value)]

    def format_all_mips(self, names: Sequence[Tuple[str, str, int, int]], middle: bytes, first_index: int = 0) -> Tuple[bytes, int]:
        """all_mips records of the window replayed last, formatted on the device: (text, number of records)."""
        arr = (RecordNames * max(len(names), 1))()
        keep = []
        for i, (chrom, label, fs, fe) in enumerate(names):
            a, b = chrom.encode(), label.encode()
            keep += [a, b]
            arr[i] = RecordNames(a, b, fs, fe)
        nrec, nb = C.c_int64(), C.c_int64()
        self._check(self.lib.mipgen_accel_format_all_mips(self.h, arr, middle, first_index, C.byref(nrec), C.byref(nb)))
        buf = C.create_string_buffer(max(nb.value, 1))
        self._check(self.lib.mipgen_accel_download_text(self.h, buf, nb.value))
        return buf.raw[:nb.value], nrec.value

    def collapse(self) -> None:
        self._check(self.lib.mipgen_accel_collapse(self.h))

    def region_bases(self, region: int) -> Tuple[int, int]:
        fe, nb = C.c_int64(), C.c_int32()
        self._check(self.lib.mipgen_accel_region_bases(self.h, region, C.byref(fe), C.byref(nb)))
        return fe.value, nb.value

    def download_collapsed(self, window: int = -1) -> np.ndarray:
        """collapse_mips result of a window (or of the whole batch, window = -1): 2 entries per base, region after region."""
        if window < 0:
            fe, nb = self.region_bases(len(self.grids) - 1) if self.grids else (0, 0)
            n = fe + 2 * nb
        else:
            wi = self.window_info(window)
            f0, _ = self.region_bases(wi["first_region"])
            f1, nb = self.region_bases(wi["first_region"] + wi["n_regions"] - 1)
            n = f1 + 2 * nb - f0
        out = np.empty(max(n, 1), dtype=np.int32)
        self._check(self.lib.mipgen_accel_download_collapsed(self.h, window, out.ctypes.data_as(C.POINTER(C.c_int32)), out.shape[0]))
        return out[:n]

    def rescore_survivors(self, window: Optional[int] = None) -> np.ndarray:
        """SVR score of every condensed survivor of the window scored + replayed last (NaN where a slot holds no survivor)."""
        self._check(self.lib.mipgen_accel_rescore_survivors(self.h))
        w = 0 if window is None else window
        n = 2 * self.window_info(w)["n_positions"]
        out = np.empty(max(n, 1), dtype=np.float64)
        self._check(self.lib.mipgen_accel_download_survivor_scores(self.h, w, out.ctypes.data_as(C.POINTER(C.c_double)), max(n, 1)))
        return out[:n]

    def window_views(self, window: int) -> "WindowViews":
        v = WindowViews()
        self._check(self.lib.mipgen_accel_window_views(self.h, window, C.byref(v)))
        return v

    def synchronize(self) -> None:
        self._check(self.lib.mipgen_accel_synchronize(self.h))

    def survivors_device_ptr(self) -> Tuple[int, int]:
        p, n = C.c_void_p(), C.c_int64()
        self._check(self.lib.mipgen_accel_survivors_device_ptr(self.h, C.byref(p), C.byref(n)))
        return p.value or 0, n.value

    def set_logistic_subruns(self, n: int) -> None:
        self._check(self.lib.mipgen_accel_set_logistic_subruns(self.h, n))

    def set_print_exact(self, on: bool) -> None:
        self._check(self.lib.mipgen_accel_set_print_exact(self.h, int(on)))

    def set_sv_split(self, n_split: int) -> None:
        self._check(self.lib.mipgen_accel_set_sv_split(self.h, n_split))

    def result_device_ptrs(self) -> Tuple[int, int]:
        a, b = C.c_void_p(), C.c_void_p()
        self._check(self.lib.mipgen_accel_result_device_ptrs(self.h, C.byref(a), C.byref(b)))
        return a.value or 0, b.value or 0

    def download(self, first: int = 0, count: Optional[int] = None) -> Tuple[np.ndarray, np.ndarray]:
        """Dense results of the window scored last ([first, first+count) are batch-wide candidate indices inside it)."""
        n = self.batch_candidates() - first if count is None else count
        scores = np.empty(n, dtype=np.float64)
        records = np.empty(n, dtype=np.uint64)
        self._check(self.lib.mipgen_accel_download_results(self.h, scores.ctypes.data_as(C.POINTER(C.c_double)),
                                                           records.ctypes.data_as(C.POINTER(C.c_uint64)), first, n))
        return scores, records

    def score_regions(self, regions: Sequence[RegionData], method: int) -> Tuple[List[Grid], np.ndarray, np.ndarray]:
        """Upload, score every result window, dense results of the whole batch to the host."""
        self.upload(regions)
        total = self.batch_candidates()
        scores = np.empty(total, dtype=np.float64)
        records = np.empty(total, dtype=np.uint64)
        for w in range(self.window_count()):
            wi = self.window_info(w)
            self.score_window(w, method)
            c0, n = wi["first_candidate"], wi["n_candidates"]
            scores[c0:c0 + n], records[c0:c0 + n] = self.download(c0, n)
        return self.grids, scores, records

    def score_regions_one_call(self, regions: Sequence[RegionData], method: int, capacity: int):
        """mipgen_accel_score_regions itself (upload + score + download behind one C call)."""
        arr = region_array(regions)
        grids = (Grid * len(regions))()
        scores = np.empty(capacity, dtype=np.float64)
        records = np.empty(capacity, dtype=np.uint64)
        self._check(self.lib.mipgen_accel_score_regions(self.h, arr, len(regions), method, grids, scores.ctypes.data_as(C.POINTER(C.c_double)),
                                                        records.ctypes.data_as(C.POINTER(C.c_uint64)), capacity))
        self.grids = list(grids)
        self._regions = regions
        n = self.batch_candidates()
        return self.grids, scores[:n], records[:n]

    def score_candidates(self, cands: Sequence[Tuple[int, int, int, int, int, int]], method: int,
                         want_features: bool = False, want_ints: bool = False):
        n = len(cands)
        arr = (Candidate * n)()
        for i, c in enumerate(cands):
            arr[i] = Candidate(*c)
        scores = np.empty(n, dtype=np.float64)
        records = np.empty(n, dtype=np.uint64)
        feats = np.empty((n, N_FEATURES), dtype=np.float64) if want_features else None
        ints = (CandidateInts * n)() if want_ints else None
        self._check(self.lib.mipgen_accel_score_candidates(
            self.h, arr, n, method, scores.ctypes.data_as(C.POINTER(C.c_double)),
            records.ctypes.data_as(C.POINTER(C.c_uint64)),
            feats.ctypes.data_as(C.POINTER(C.c_double)) if feats is not None else None,
            ints if ints is not None else None))
        return scores, records, feats, ints

    def score_candidate_array(self, arr, n: int, method: int) -> np.ndarray:
        """score_candidates() for a ready-made ctypes array of Candidate: scores only."""
        scores = np.empty(max(n, 1), dtype=np.float64)
        if n:
            self._check(self.lib.mipgen_accel_score_candidates(self.h, arr, n, method, scores.ctypes.data_as(C.POINTER(C.c_double)), None, None, None))
        return scores[:n]

    def long_range_content_batch(self, seqs: Sequence[bytes], starts: Sequence[int], stops: Sequence[int]) -> np.ndarray:
        n = len(seqs)
        out = np.empty((n, N_LRC), dtype=np.float64)
        if n == 0:
            return out
        arr = (C.c_char_p * n)(*seqs)
        lens = np.array([len(s) for s in seqs], dtype=np.int32)
        st = np.ascontiguousarray(starts, dtype=np.int32)
        sp = np.ascontiguousarray(stops, dtype=np.int32)
        i32p = C.POINTER(C.c_int32)
        self._check(self.lib.mipgen_accel_long_range_content_batch(self.h, n, arr, lens.ctypes.data_as(i32p), st.ctypes.data_as(i32p),
                                                                   sp.ctypes.data_as(i32p), out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def long_range_content(self, extended_seq: bytes, chrom_seq_start: int, chrom_seq_stop: int) -> np.ndarray:
        out = np.empty(N_LRC, dtype=np.float64)
        self._check(self.lib.mipgen_accel_long_range_content(self.h, extended_seq, len(extended_seq), chrom_seq_start,
                                                             chrom_seq_stop, out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def format_all_mips_array(self, names_arr, middle: bytes, first_index: int = 0) -> Tuple[bytes, int]:
        """format_all_mips() for a ready-made ctypes array of RecordNames (hostapi.Design.record_names)."""
        nrec, nb = C.c_int64(), C.c_int64()
        self._check(self.lib.mipgen_accel_format_all_mips(self.h, names_arr, middle, first_index, C.byref(nrec), C.byref(nb)))
        buf = C.create_string_buffer(max(nb.value, 1))
        self._check(self.lib.mipgen_accel_download_text(self.h, buf, nb.value))
        return buf.raw[:nb.value], nrec.value

    def replay_condense(self) -> None:
        self._check(self.lib.mipgen_accel_replay_condense(self.h))

    def download_replay(self, want_mask: bool = True, window: Optional[int] = None):
        """Replay / condense results of the window replayed last (a single-window batch: the whole batch)."""
        if window is None and self.window_count() == 1:
            nreg, npos, total = len(self.grids), sum(g.n_pos for g in self.grids), self.batch_candidates()
        else:
            wi = self.window_info(window if window is not None else 0)
            nreg, npos, total = wi["n_regions"], wi["n_positions"], wi["n_candidates"]
        emitted = np.zeros(nreg, dtype=np.int64)
        surv = np.zeros(2 * npos, dtype=SURVIVOR_DTYPE)
        mask = np.zeros(total if want_mask else 0, dtype=np.uint8)
        self._check(self.lib.mipgen_accel_download_replay(
            self.h, emitted.ctypes.data_as(C.POINTER(C.c_int64)), surv.ctypes.data_as(C.POINTER(Survivor)), 2 * npos,
            mask.ctypes.data_as(C.POINTER(C.c_uint8)) if want_mask else None, mask.shape[0]))
        return emitted, surv, mask

    def set_timing(self, on: bool) -> None:
        self._check(self.lib.mipgen_accel_set_timing(self.h, int(on)))

    def last_kernel_ms(self, which: int = 0) -> float:
        return float(self.lib.mipgen_accel_last_kernel_ms(self.h, which))


# record field accessors (vectorised)
def rec_ext_copy(r): return (r & np.uint64(0xFFFF)).astype(np.int64)
def rec_lig_copy(r): return ((r >> np.uint64(16)) & np.uint64(0xFFFF)).astype(np.int64)
def rec_masked_n(r): return ((r >> np.uint64(32)) & np.uint64(0xFF)).astype(np.int64)
def rec_snp_count(r): return ((r >> np.uint64(40)) & np.uint64(0xFF)).astype(np.int64)
def rec_flags(r): return ((r >> np.uint64(48)) & np.uint64(0xFF)).astype(np.int64)
def rec_junction(r): return ((r >> np.uint64(56)) & np.uint64(0xFF)).astype(np.int64)

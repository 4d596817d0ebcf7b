"""ctypes binding of include/mipgen_host.h (libmipgen_host.so: options + input stage + selection stage of the drop-in front end).

Plumbing for tests and bench.py: with it the torch.distributed harness can end where the reference does - rank 0 feeds the gathered
survivors through the sequential pick stage and writes picked_mips.txt.  The host library is C++ (mipgen_amd/host/); nothing is
re-implemented here.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Callable, List, Optional, Sequence

import numpy as np

from . import capi

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libmipgen_host.so")

RESCORE_FN = C.CFUNCTYPE(C.c_double, C.c_void_p, C.c_int32, C.POINTER(capi.Candidate))
# numpy image of capi.Grid (mipgen_grid)
GRID_DTYPE = np.dtype([(name, np.int64 if ct is C.c_int64 else np.int32) for name, ct in capi.Grid._fields_], align=True)
assert GRID_DTYPE.itemsize == C.sizeof(capi.Grid)

EXPORTED_SYMBOLS = [
    "mipgen_host_last_error", "mipgen_host_last_circumstance", "mipgen_design_open", "mipgen_design_close", "mipgen_design_params",
    "mipgen_design_score_method", "mipgen_design_silent", "mipgen_design_model_path", "mipgen_design_region_count", "mipgen_design_region", "mipgen_design_regions",
    "mipgen_design_long_range_seq", "mipgen_design_set_long_range_content", "mipgen_design_select_region",
    "mipgen_design_select_region_collapsed", "mipgen_design_select_regions", "mipgen_design_survivor_candidates", "mipgen_design_record_names", "mipgen_design_middle",
    "mipgen_design_write_all_mips", "mipgen_design_counters", "mipgen_design_region_weights",
    "mipgen_design_run", "mipgen_design_set_devices", "mipgen_design_set_api_device", "mipgen_design_set_window_candidates", "mipgen_design_set_timing", "mipgen_design_set_gather", "mipgen_host_rand_stream",
]

_lib = None


class HostError(RuntimeError):
    pass


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HostError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
    capi.load_library()                                   # libmipgen_accel.so first (rpath $ORIGIN resolves it as well)
    lib = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    lib.mipgen_host_last_error.restype = C.c_char_p
    lib.mipgen_design_open.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(vp)]
    lib.mipgen_design_close.argtypes = [vp]
    lib.mipgen_design_params.argtypes = [vp, C.POINTER(capi.Params)]
    lib.mipgen_design_score_method.argtypes = [vp]
    lib.mipgen_design_silent.argtypes = [vp]
    lib.mipgen_design_model_path.argtypes = [vp]
    lib.mipgen_design_model_path.restype = C.c_char_p
    lib.mipgen_design_region_count.argtypes = [vp]
    lib.mipgen_design_region.argtypes = [vp, C.c_int32, C.POINTER(capi.Region)]
    lib.mipgen_design_regions.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(capi.Region)]
    lib.mipgen_design_long_range_seq.argtypes = [vp, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_int32)]
    lib.mipgen_design_set_long_range_content.argtypes = [vp, C.c_int32, C.POINTER(C.c_double)]
    lib.mipgen_design_select_region.argtypes = [vp, C.c_int32, C.POINTER(capi.Grid), C.POINTER(capi.Survivor), C.c_int64, C.POINTER(C.c_double),
                                                C.POINTER(C.c_uint64), C.POINTER(C.c_uint8), RESCORE_FN, vp]
    lib.mipgen_design_select_region_collapsed.argtypes = [vp, C.c_int32, C.POINTER(capi.Grid), C.POINTER(capi.Survivor), C.c_int64, C.POINTER(C.c_double),
                                                          C.POINTER(C.c_uint64), C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.c_int32, RESCORE_FN, vp]
    lib.mipgen_design_counters.argtypes = [vp] + [C.POINTER(C.c_int64)] * 4
    lib.mipgen_design_region_weights.argtypes = [vp, C.POINTER(C.c_int64), C.c_int32]
    lib.mipgen_design_select_regions.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(capi.Grid), C.POINTER(capi.Survivor), C.POINTER(C.c_int64),
                                                 C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_double)]
    lib.mipgen_design_record_names.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(capi.RecordNames)]
    lib.mipgen_design_middle.argtypes = [vp]
    lib.mipgen_design_middle.restype = C.c_char_p
    lib.mipgen_design_write_all_mips.argtypes = [vp, C.c_char_p, C.c_int64, C.c_int64]
    lib.mipgen_design_survivor_candidates.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(capi.Grid), C.POINTER(capi.Survivor), C.POINTER(capi.Candidate),
                                                      C.POINTER(C.c_int64), C.c_int64, C.POINTER(C.c_int64)]
    lib.mipgen_design_run.argtypes = [vp, C.c_int32]
    lib.mipgen_design_set_devices.argtypes = [vp, C.c_int32]
    lib.mipgen_design_set_api_device.argtypes = [vp, C.c_int32]
    lib.mipgen_design_set_window_candidates.argtypes = [vp, C.c_int64]
    lib.mipgen_design_set_timing.argtypes = [vp, C.c_int32]
    lib.mipgen_design_set_gather.argtypes = [vp, C.c_int32]
    lib.mipgen_host_rand_stream.argtypes = [C.POINTER(C.c_int32), C.c_int32]
    _lib = lib
    return lib


class Design:
    """A mipgen design opened from a command line: options parsed, input stage done, output files open."""

    def __init__(self, argv: Sequence[str]):
        self.lib = load_library()
        arr = (C.c_char_p * len(argv))(*[a.encode() for a in argv])
        self.h = C.c_void_p()
        rc = self.lib.mipgen_design_open(len(argv), arr, C.byref(self.h))
        if rc:
            raise HostError(f"mipgen_design_open: {rc}: {self.lib.mipgen_host_last_error().decode()}")
        self._keep: List[object] = []

    def _check(self, rc: int) -> None:
        if rc:
            raise HostError(f"mipgen_host error {rc}: {self.lib.mipgen_host_last_error().decode()}")

    def close(self) -> None:
        if self.h:
            self.lib.mipgen_design_close(self.h)
            self.h = C.c_void_p()

    def params(self) -> capi.Params:
        p = capi.Params()
        self._check(self.lib.mipgen_design_params(self.h, C.byref(p)))
        return p

    @property
    def score_method(self) -> int:
        return int(self.lib.mipgen_design_score_method(self.h))

    @property
    def silent(self) -> bool:
        return bool(self.lib.mipgen_design_silent(self.h))

    @property
    def model_path(self) -> str:
        return self.lib.mipgen_design_model_path(self.h).decode()

    def set_api_device(self, device: int) -> None:
        self._check(self.lib.mipgen_design_set_api_device(self.h, device))

    def region_count(self) -> int:
        return int(self.lib.mipgen_design_region_count(self.h))

    def region(self, i: int) -> capi.Region:
        """The C view of region i (pointers into the design's host arrays; valid until close)."""
        r = capi.Region()
        self._check(self.lib.mipgen_design_region(self.h, i, C.byref(r)))
        return r

    def regions(self, first: int, n: int):
        """The C views of regions first .. first + n - 1 as one ctypes array (what mipgen_accel_upload_regions takes)."""
        arr = (capi.Region * max(n, 1))()
        self._check(self.lib.mipgen_design_regions(self.h, first, n, arr))
        return arr

    def long_range_seq(self, i: int) -> bytes:
        s, n = C.c_char_p(), C.c_int32()
        self._check(self.lib.mipgen_design_long_range_seq(self.h, i, C.byref(s), C.byref(n)))
        return C.string_at(s, n.value) if n.value else b""

    def set_long_range_content(self, i: int, lrc: np.ndarray) -> None:
        lrc = np.ascontiguousarray(lrc, dtype=np.float64)
        self._check(self.lib.mipgen_design_set_long_range_content(self.h, i, lrc.ctypes.data_as(C.POINTER(C.c_double))))

    def select_region(self, i: int, grid: capi.Grid, survivors: np.ndarray, emitted: int, scores: Optional[np.ndarray] = None,
                      records: Optional[np.ndarray] = None, mask: Optional[np.ndarray] = None,
                      rescore: Optional[Callable[[int, capi.Candidate], float]] = None, collapsed: Optional[np.ndarray] = None) -> None:
        survivors = np.ascontiguousarray(survivors)
        assert survivors.dtype == capi.SURVIVOR_DTYPE and survivors.shape[0] == 2 * grid.n_pos
        fn = RESCORE_FN(lambda ctx, region, cand: float(rescore(region, cand.contents))) if rescore else RESCORE_FN()
        dp = C.POINTER(C.c_double)
        args = (self.h, i, C.byref(grid), survivors.ctypes.data_as(C.POINTER(capi.Survivor)), emitted,
                scores.ctypes.data_as(dp) if scores is not None else None,
                records.ctypes.data_as(C.POINTER(C.c_uint64)) if records is not None else None,
                mask.ctypes.data_as(C.POINTER(C.c_uint8)) if mask is not None else None)
        if collapsed is not None:
            collapsed = np.ascontiguousarray(collapsed, dtype=np.int32)
            self._check(self.lib.mipgen_design_select_region_collapsed(*args, collapsed.ctypes.data_as(C.POINTER(C.c_int32)), collapsed.shape[0] // 2, fn, None))
        else:
            self._check(self.lib.mipgen_design_select_region(*args, fn, None))

    @staticmethod
    def _grid_array(grids: np.ndarray) -> np.ndarray:
        g6 = np.ascontiguousarray(grids, dtype=np.int64).reshape(-1, 6)
        garr = np.zeros(g6.shape[0], dtype=GRID_DTYPE)
        for k, f in enumerate(("offset", "count", "first_pos", "n_pos", "first_size_index", "n_sizes")):
            garr[f] = g6[:, k]
        return garr

    def select_regions(self, first: int, grids: np.ndarray, survivors: np.ndarray, emitted: np.ndarray, collapsed: Optional[np.ndarray] = None,
                       n_bases: Optional[np.ndarray] = None, svr: Optional[np.ndarray] = None) -> None:
        """Selection stage for a run of regions of a silent design in one call: `grids` an int64 array [n][6] (offset, count, first_pos, n_pos,
        first_size_index, n_sizes per region), `survivors` the regions' survivors one after the other, `emitted` their emitted counts; `collapsed` /
        `n_bases`: the accelerator's collapse results of the regions (capi.Accel.download_collapsed) and their bases per region; `svr` (mixed designs):
        the SVR score of every survivor, parallel to `survivors`."""
        survivors = np.ascontiguousarray(survivors)
        emitted = np.ascontiguousarray(emitted, dtype=np.int64)
        garr = self._grid_array(grids)
        n = garr.shape[0]
        assert survivors.dtype == capi.SURVIVOR_DTYPE and emitted.shape[0] == n and survivors.shape[0] == 2 * int(garr["n_pos"].astype(np.int64).sum())
        i32p = C.POINTER(C.c_int32)
        cp = nbp = sp = None
        if collapsed is not None:
            collapsed = np.ascontiguousarray(collapsed, dtype=np.int32)
            n_bases = np.ascontiguousarray(n_bases, dtype=np.int32)
            assert n_bases.shape[0] == n and collapsed.shape[0] == 2 * int(n_bases.astype(np.int64).sum())
            cp, nbp = collapsed.ctypes.data_as(i32p), n_bases.ctypes.data_as(i32p)
        if svr is not None:
            svr = np.ascontiguousarray(svr, dtype=np.float64)
            assert svr.shape[0] == survivors.shape[0]
            sp = svr.ctypes.data_as(C.POINTER(C.c_double))
        self._check(self.lib.mipgen_design_select_regions(self.h, first, n, garr.ctypes.data_as(C.POINTER(capi.Grid)),
                                                          survivors.ctypes.data_as(C.POINTER(capi.Survivor)), emitted.ctypes.data_as(C.POINTER(C.c_int64)), cp, nbp, sp))

    def survivor_candidates(self, first: int, grids: np.ndarray, survivors: np.ndarray):
        """Mixed designs: the survivors of regions first .. as a ctypes array of capi.Candidate (region = index inside the run) for an SVR re-score on
        the handle that holds exactly these regions, and the survivor slot of every candidate."""
        survivors = np.ascontiguousarray(survivors)
        garr = self._grid_array(grids)
        n = garr.shape[0]
        m = int((survivors["cand_index"] >= 0).sum())
        cands = (capi.Candidate * max(m, 1))()
        where = np.zeros(max(m, 1), dtype=np.int64)
        cnt = C.c_int64()
        self._check(self.lib.mipgen_design_survivor_candidates(self.h, first, n, garr.ctypes.data_as(C.POINTER(capi.Grid)), survivors.ctypes.data_as(C.POINTER(capi.Survivor)),
                                                               cands, where.ctypes.data_as(C.POINTER(C.c_int64)), m, C.byref(cnt)))
        assert cnt.value == m
        return cands, where[:m], m

    def record_names(self, first: int, n: int):
        """ctypes array of capi.RecordNames for regions first .. first + n - 1 (what mipgen_accel_format_all_mips takes)."""
        arr = (capi.RecordNames * max(n, 1))()
        self._check(self.lib.mipgen_design_record_names(self.h, first, n, arr))
        return arr

    @property
    def middle(self) -> bytes:
        return self.lib.mipgen_design_middle(self.h)

    def write_all_mips(self, text: bytes, renumber_base: int = 0) -> None:
        """Append device-formatted all_mips records to the design's file, their record numbers raised by renumber_base."""
        self._check(self.lib.mipgen_design_write_all_mips(self.h, text, len(text), renumber_base))

    def region_weights(self) -> np.ndarray:
        """Relative device time per region: the weights of the device shards (mipgen_design_run's own rule)."""
        n = self.region_count()
        w = np.zeros(max(n, 1), dtype=np.int64)
        self._check(self.lib.mipgen_design_region_weights(self.h, w.ctypes.data_as(C.POINTER(C.c_int64)), n))
        return w[:n]

    def counters(self):
        a, b, c, d = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        self._check(self.lib.mipgen_design_counters(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return {"all_mips": a.value, "collapsed": b.value, "picked": c.value, "gaps": d.value}

    def run(self, n_devices: int = 0, window_candidates: int = 0, timing: bool = False) -> None:
        if window_candidates:
            self._check(self.lib.mipgen_design_set_window_candidates(self.h, window_candidates))
        if timing:
            self._check(self.lib.mipgen_design_set_timing(self.h, 1))
        self._check(self.lib.mipgen_design_run(self.h, n_devices))

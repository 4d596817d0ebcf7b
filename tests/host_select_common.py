"""Shared by tests/test_host_select_cpu.py and its gloo worker: drive the host library's selection stage (libmipgen_host.so, C++) with
survivors computed on the CPU by the oracle.  Test infrastructure."""
from __future__ import annotations

import os
from typing import Dict, List, Tuple

import numpy as np

from mipgen_amd import capi, hostapi
from oracle import pyoracle as po


class RegionView:
    """A design's region in the C layout (pointers into libmipgen_host's arrays), shaped like capi.RegionData for the oracle wrappers."""

    def __init__(self, c):
        self.c = c
        self.alleles = None


import contextlib


@contextlib.contextmanager
def in_dir(work: str):
    """The front end resolves -project_name, the TRF mask (mipgen.cpp:1051-1054) and the gap files relative to the CWD, as the reference does."""
    cwd = os.getcwd()
    os.chdir(work)
    try:
        yield
    finally:
        os.chdir(cwd)


def open_design(argv: List[str], work: str) -> hostapi.Design:
    with in_dir(work):
        return hostapi.Design(argv)


def design_views(d: hostapi.Design) -> List[RegionView]:
    """Region views with the long-range content filled in by the oracle (svr / mixed designs)."""
    views = []
    for i in range(d.region_count()):
        if d.score_method != capi.SCORE_LOGISTIC:
            r = d.region(i)
            d.set_long_range_content(i, po.long_range_content(d.long_range_seq(i), r.seq_start, r.seq_stop))
        views.append(RegionView(d.region(i)))
    return views


def oracle_region_results(P: capi.Params, view: RegionView, scan_method: int, model) -> Dict[str, object]:
    """What the accelerator hands the selection stage for one region, computed by the oracle: dense scores / records, emitted mask,
    condensed survivors (region-local candidate indices, grid offset 0)."""
    g, scores, records = po.score_region_dense(P, view, scan_method, model)
    n_emit, mask = po.replay_region(P, view, scores, records)
    surv = po.condense_region(P, view, scores, records, mask)
    return {"grid": g, "scores": scores, "records": records, "mask": mask, "survivors": surv, "emitted": n_emit}


def make_rescorer(P: capi.Params, views: List[RegionView], model):
    def rescore(region: int, cand) -> float:
        v = views[region]
        sk, dsg = po.design(P, v, (0, cand.scan_start, cand.capture_size, cand.ext_len, cand.lig_len, cand.strand))
        s, _, _ = po.score_designed(dsg, capi.SCORE_SVR, np.array(v.c.long_range_content[:]), model)
        return s
    return rescore

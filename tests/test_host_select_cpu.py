"""CPU: the host library (libmipgen_host.so: options, input stage, selection stage behind include/mipgen_host.h) fed with survivors the
ORACLE computed, against the files the real reference wrote on the same designs (tests/golden/design_*).  No GPU is involved: this pins
the C++ collapse / pick / print code on its own, and - with world size 2 over gloo - the multi-rank path: regions sharded over the
ranks, one gather of the condensed survivors, sequential pick on rank 0."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from mipgen_amd import capi, hostapi
from oracle import pyoracle as po
from tests import helpers as H
from tests import host_select_common as HS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DESIGNS = sorted(d[len("design_"):] for d in os.listdir(H.GOLDEN) if d.startswith("design_"))
# the CPU oracle needs ~10 minutes of one core for the 1.55e7-candidate SVR grid of BASELINE configs[1]: that design is checked on the GPU only
# (tests/test_gpu_cli.py), the CPU suite keeps all the others
# (the silent low-complexity SVR design of the hard genome - 7.4e5 candidates under a 200-SV model - runs through the whole command line on the
# oracle-backed stub in tests/test_host_threads_cpu.py instead)
HEAVY = {"practice62_config2_svr", "hard_lowcomplexity_svr_silent", "exome1000_logistic_silent", "exome1000_mixed_silent"}   # (the last two: 1.4e8 candidates each - GPU only)
DESIGNS = [d for d in DESIGNS if d not in HEAVY]


def test_host_library_exports_every_declared_symbol():
    lib = hostapi.load_library()
    for sym in hostapi.EXPORTED_SYMBOLS:
        getattr(lib, sym)
    hdr = open(os.path.join(ROOT, "include", "mipgen_host.h")).read()
    import re
    declared = set(re.findall(r"\b(mipgen_(?:design|host)_[a-z_]+)\s*\(", hdr))
    assert declared == set(hostapi.EXPORTED_SYMBOLS), declared ^ set(hostapi.EXPORTED_SYMBOLS)


def test_private_rand_stream_is_glibc_rand():
    """The reference orders the two strands with a never-seeded libc rand() (mipgen.cpp:1863); the selection stage carries its own copy
    of that generator (nothing else in the process can advance it): 20,000 values equal a fresh process's rand()."""
    import ctypes as C
    n = 20000
    buf = (C.c_int32 * n)()
    hostapi.load_library().mipgen_host_rand_stream(buf, n)
    ref = subprocess.run([sys.executable, "-c", f"import ctypes; l = ctypes.CDLL('libc.so.6'); print(' '.join(str(l.rand()) for _ in range({n})))"],
                         stdout=subprocess.PIPE, check=True).stdout.split()
    assert [int(x) for x in ref] == list(buf)


@pytest.mark.parametrize("name", DESIGNS)
def test_selection_stage_on_oracle_survivors_matches_reference_files(name, tmp_path):
    meta = H.load_design(name)
    work = str(tmp_path)
    argv = H.prepare_cli_workdir(meta, work)
    env_mode = os.environ.get("FAKEBWA_MODE")
    os.environ["FAKEBWA_MODE"] = meta["bwa"]
    try:
        d = HS.open_design(argv, work)
    finally:
        if env_mode is None:
            del os.environ["FAKEBWA_MODE"]
        else:
            os.environ["FAKEBWA_MODE"] = env_mode
    P = d.params()
    model = po.Model(d.model_path) if d.score_method != capi.SCORE_LOGISTIC else None
    views = HS.design_views(d)
    scan = capi.SCORE_SVR if d.score_method == capi.SCORE_SVR else capi.SCORE_LOGISTIC
    rescore = HS.make_rescorer(P, views, model) if d.score_method == capi.SCORE_MIXED else None
    with HS.in_dir(work):
        for i, v in enumerate(views):
            r = HS.oracle_region_results(P, v, scan, model)
            d.select_region(i, r["grid"], r["survivors"], r["emitted"], r["scores"], r["records"], r["mask"], rescore)
    with pytest.raises(hostapi.HostError):
        d.select_region(0, r["grid"], r["survivors"], r["emitted"])          # out of order
    assert d.counters()["all_mips"] == meta["lines"]["all_mips"] - 1
    assert d.counters()["picked"] == meta["lines"]["picked_mips"] - 1
    d.close()
    H.compare_outputs(meta, work)


def test_two_ranks_gather_survivors_and_rank0_picks(tmp_path):
    """world size 2 over gloo: each rank scores its contiguous shard of the design's regions (oracle on CPU), the condensed survivors go
    to rank 0 in ONE gather, rank 0 runs the sequential selection stage over all regions: picked / collapsed / snp files equal the
    reference's single-process files."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = tmp_path / "r0.json"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "host_select_worker.py"), "logistic_default_arms", str(tmp_path), str(out)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert p.returncode == 0, p.stdout.decode()[-3000:]
    res = json.load(open(out))
    assert res["shards"][0][1] == res["shards"][1][0] and res["shards"][1][1] == res["n_regions"] and res["shards"][0][1] > 0
    meta = H.load_design("logistic_default_arms")
    assert res["picked"] == meta["lines"]["picked_mips"] - 1
    H.compare_outputs(meta, os.path.join(str(tmp_path), "rank0"), keys=("picked_mips", "snp_mips"), check_all=False)      # -silent_mode: no collapsed file

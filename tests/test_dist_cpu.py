"""CPU: the N>1 host path (region sharding + one gather of survivors to rank 0) with world size 2 over gloo."""
import json
import os
import socket
import subprocess
import sys

import numpy as np

from mipgen_amd import dist as mdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_regions_is_contiguous_balanced_and_complete():
    rng = np.random.default_rng(1)
    for world in (1, 2, 3, 4, 8):
        for n in (1, 2, 7, 8, 62, 200):
            w = rng.integers(1, 10_000, size=n).tolist()
            sh = mdist.shard_regions(w, world)
            assert len(sh) == world
            assert sh[0][0] == 0 and sh[-1][1] == n
            assert all(sh[i][1] == sh[i + 1][0] for i in range(world - 1))
            if n >= world:
                assert all(hi > lo for lo, hi in sh)          # nobody idles when there is enough work
                loads = [sum(w[lo:hi]) for lo, hi in sh]
                assert max(loads) <= sum(w) / world + max(w)  # within one region of the ideal


def test_two_rank_gather_over_gloo(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = tmp_path / "r0.json"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py"), str(out)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert p.returncode == 0, p.stdout.decode()[-2000:]
    res = json.load(open(out))
    assert res["ok"] and res["total"] == res["expected_total"]
    assert len(res["shards"]) == 2 and res["shards"][0][1] == res["shards"][1][0]


def test_region_cost_weights_and_contiguous_shards():
    """Shard weights follow the kernels' cost model (mipgen_amd/dist.py: region_cost = design.cpp: region_cost): for SVR designs a region with few
    capture sizes costs more per candidate than its dense-grid size says; logistic weights are the dense counts; shards stay contiguous and cover
    every region once."""
    import numpy as np
    from mipgen_amd import capi, dist as mdist, workloads
    P = capi.make_params(150, 170, score_method=capi.SCORE_SVR)
    _, ivs = workloads.exome_layout()
    ivs = ivs[:3000]
    dense = workloads.dense_candidates(ivs, P)
    w_svr = workloads.shard_weights(ivs, P, True)
    w_log = workloads.shard_weights(ivs, P, False)
    assert np.array_equal(w_log, dense.astype(np.float64))
    cand, n_pos, n_sizes = workloads.dense_candidates(ivs, P, detail=True)
    per_cand = w_svr / np.maximum(cand, 1)
    few, many = per_cand[(n_sizes == 1) & (cand > 0)], per_cand[(n_sizes == n_sizes.max()) & (cand > 0)]
    assert few.size and many.size and few.min() > many.max()            # K = 1 exons: more table entries per candidate
    for world in (2, 3, 8):
        sh = mdist.shard_regions(w_svr.tolist(), world)
        assert sh[0][0] == 0 and sh[-1][1] == len(ivs) and all(a[1] == b[0] for a, b in zip(sh, sh[1:])) and all(hi > lo for lo, hi in sh)
        tot = [w_svr[lo:hi].sum() for lo, hi in sh]
        assert max(tot) / (sum(tot) / world) < 1.05


def test_multi_process_design_has_no_cpu_path():
    """mipgen_amd/mp_design.py (one process per GPU) refuses to run without a HIP device - the product never falls back to the CPU - and says what
    it wants when the mipgen flags are missing."""
    import sys
    import torch
    env = dict(os.environ, PYTHONPATH=ROOT)
    p = subprocess.run([sys.executable, "-m", "mipgen_amd.mp_design", "--gpus", "1"], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode != 0 and b"the mipgen flags follow" in p.stderr
    if torch.cuda.is_available():
        return
    p = subprocess.run([sys.executable, "-m", "mipgen_amd.mp_design", "--gpus", "1", "--", "-regions_to_scan", "x.bed"], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode != 0 and b"no CPU fallback" in p.stderr

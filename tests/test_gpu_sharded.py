"""GPU: the multi-GPU design flow on one device - the regions of a design sharded over two accelerator handles exactly as
mipgen_amd/dist.py shards them over ranks, each shard scored + replayed + condensed on its own handle (silent path), the survivors
concatenated in rank order (what the RCCL gather delivers) and fed through the sequential selection stage of libmipgen_host.so:
the picked / snp files equal the single-handle run AND the files the real reference wrote."""
import os

import numpy as np
import pytest

from mipgen_amd import capi, dist as mdist, hostapi
from tests import helpers as H
from tests import host_select_common as HS

pytestmark = pytest.mark.gpu


def _free_port() -> int:
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run(meta, work, n_shards):
    os.makedirs(work)
    os.environ["FAKEBWA_MODE"] = meta["bwa"]
    d = HS.open_design(H.prepare_cli_workdir(meta, work) + ["-silent_mode", "on"], work)
    P = d.params()
    n = d.region_count()
    scan = capi.SCORE_SVR if d.score_method == capi.SCORE_SVR else capi.SCORE_LOGISTIC
    # long-range content on the device, through the host library's own region views
    if d.score_method != capi.SCORE_LOGISTIC:
        acc0 = capi.Accel(P)
        views = [d.region(i) for i in range(n)]
        lrc = acc0.long_range_content_batch([d.long_range_seq(i) for i in range(n)], [v.seq_start for v in views], [v.seq_stop for v in views])
        for i in range(n):
            d.set_long_range_content(i, lrc[i])
        acc0.close()
    views = [HS.RegionView(d.region(i)) for i in range(n)]
    probe = capi.Accel(P)
    weights = [g.count for g in probe.upload(views)]
    probe.close()
    shards = mdist.shard_regions(weights, n_shards)
    surv_parts, emitted_parts, grids = [], [], []
    handles = []
    for lo, hi in shards:                                       # one handle per "rank"
        acc = capi.Accel(P)
        if d.score_method != capi.SCORE_LOGISTIC:
            acc.load_model_file(d.model_path)
        acc.set_sv_split(1)                                     # same summation order whatever the shard size: bitwise comparable
        g = acc.upload(views[lo:hi])
        acc.score_condense_all(scan)
        e, s = acc.download_survivors()
        for gi in g:
            gi2 = capi.Grid(); gi2.offset, gi2.count, gi2.first_pos, gi2.n_pos, gi2.first_size_index, gi2.n_sizes = gi.offset, gi.count, gi.first_pos, gi.n_pos, gi.first_size_index, gi.n_sizes
            grids.append(gi2)
        surv_parts.append(s); emitted_parts.append(e)
        handles.append((acc, lo))
    surv = np.concatenate(surv_parts)
    emitted = np.concatenate(emitted_parts)
    model_acc = {lo: acc for acc, lo in handles}

    def rescore(region, cand):                                  # mixed designs: rank 0's local accelerator re-scores
        k = max(i for i, (lo, hi) in enumerate(shards) if lo <= region)
        acc, lo = handles[k]
        s, _, _, _ = acc.score_candidates([(region - lo, cand.scan_start, cand.capture_size, cand.ext_len, cand.lig_len, cand.strand)], capi.SCORE_SVR)
        return float(s[0])
    pos = 0
    with HS.in_dir(work):
        for i, g in enumerate(grids):
            d.select_region(i, g, surv[2 * pos:2 * (pos + g.n_pos)], int(emitted[i]), rescore=rescore if d.score_method == capi.SCORE_MIXED else None)
            pos += g.n_pos
    c = d.counters()
    d.close()
    for acc, _ in handles:
        acc.close()
    return surv, emitted, c, shards


@pytest.mark.parametrize("name", ["long_default", "mixed_12_regions", "svr_2kb", "double_tile_both"])
def test_sharded_survivors_through_rank0_pick(name, tmp_path):
    meta = H.load_design(name)
    s1, e1, c1, _ = _run(meta, str(tmp_path / "one"), 1)
    n_sh = 2 if len(meta["intervals"]) > 1 else 1
    s2, e2, c2, shards = _run(meta, str(tmp_path / "two"), n_sh)
    assert np.array_equal(e1, e2) and c1 == c2
    assert np.array_equal(s1["record"], s2["record"])
    assert np.array_equal(s1["score"], s2["score"], equal_nan=True)          # same kernels, same summation order: bit-identical
    # candidate indices are batch-wide: compare them per region (the second shard restarts at 0)
    for w in ("one", "two"):
        H.compare_outputs(meta, str(tmp_path / w), keys=("picked_mips", "snp_mips"), check_all=False)
    assert c1["picked"] == meta["lines"]["picked_mips"] - 1 and c1["all_mips"] == meta["lines"]["all_mips"] - 1


def test_bench_two_ranks_child_process():
    """bench.py --gpus 2: the launcher path the driver's SCALE run takes (one process per GPU over RCCL, the headline batch weak-scaled to two instances
    and cut by the region cost model, one gather of the condensed survivors per step; the exome cut two ways beside it as `exome_strong`).  Runs as a
    CHILD process - a GPU-initialised process never execs - and only where two devices are visible."""
    import json
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--scale-base-regions", "2048", "--no-cpu-baseline"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, cwd=root)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    line = json.loads(p.stdout.decode().strip().splitlines()[-1])
    _check_two_rank_line(line, "nccl")


def _check_two_rank_line(line, backend):
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["backend"] == backend
    assert line["scaling"] == "weak" and "practice62 x2" in line["config"]["workload"] and "weak" in line["scale_family"]
    assert line["parity_checked"] is True and line["value"] > 0
    # both ranks hold a share of the BED (cost-model shards: within a factor of 1.5 of each other), and rank 1's survivors arrived too
    per = line["config"]["dense_candidates_per_rank"]
    assert len(per) == 2 and min(per) > 0 and sum(per) == line["config"]["dense_candidates_all_ranks"] and max(per) < 1.5 * min(per)
    assert line["config"]["survivors_gathered_per_step"] > line["config"]["survivors_rank0"] > 0
    # the metric's own multi-GPU config beside it: the same exome BED cut two ways, one timed pass with its gather
    ex = line["exome_strong"]
    assert ex["n_gpus"] == 2 and ex["scaling"] == "strong" and "first 2048" in ex["what"] and ex["value"] > 0
    assert len(ex["dense_candidates_per_rank"]) == 2 and sum(ex["dense_candidates_per_rank"]) == ex["dense_candidates"]
    assert ex["survivors_gathered"] > ex["survivors_rank0"] > 0


def test_bench_two_ranks_logic_through_gloo_on_one_gpu():
    """The N > 1 path of bench.py on a box with ONE GPU: two ranks share the device and exchange through host memory (`--backend gloo --share-gpus`,
    tests only - never a measurement).  Everything but RCCL itself is the code the driver's SCALE run executes: the launcher, the cost-model shards,
    the size exchange, the per-step gather of the condensed survivors to rank 0, the max-over-ranks clock, the per-rank candidate counts - for the
    weak-scaled headline family and for the exome cut two ways (`exome_strong`)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpus", "--steps", "2", "--warmup", "1",
                        "--scale-base-regions", "2048", "--exome-regions", "0", "--sustain-seconds", "0", "--no-cpu-baseline"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, cwd=root)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    line = json.loads(p.stdout.decode().strip().splitlines()[-1])
    assert line["shared_gpus"] is True
    _check_two_rank_line(line, "gloo")
    # the same exome BED on one rank: the same dense-candidate total
    q = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--config", "exome", "--regions", "2048", "--scaling", "strong", "--steps", "1",
                        "--warmup", "1", "--no-cpu-baseline", "--no-extras", "--no-parity-gate"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, cwd=root)
    assert q.returncode == 0, q.stderr.decode()[-2000:]
    one = json.loads(q.stdout.decode().strip().splitlines()[-1])
    assert one["config"]["dense_candidates_all_ranks"] == line["exome_strong"]["dense_candidates"]
    # an explicit strong-scaling run of the exome family (one BED cut two ways in `value` itself) still works
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpus", "--config", "exome", "--scaling", "strong",
                        "--regions", "1024", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-extras"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    two = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert two["scaling"] == "strong" and "first 1024" in two["config"]["workload"] and len(two["config"]["dense_candidates_per_rank"]) == 2


def test_bench_scale_base_line_is_the_sharded_workload():
    """The N = 1 point of the scaling curve: `bench.py --gpus 1 --config exome --regions R --scaling strong` runs, on one GPU, exactly the BED
    that `--gpus N` cuts N ways (same describe string, same dense-candidate total as the sum over the ranks of an N-rank run), and names one
    RCCL-free rank."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--config", "exome", "--regions", "2048", "--scaling", "strong",
                        "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, cwd=root)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    line = json.loads(p.stdout.decode().strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["rccl_ranks"] == 1 and line["scaling"] == "strong"
    assert "exome200k" in line["config"]["workload"] and "first 2048" in line["config"]["workload"]
    assert line["config"]["dense_candidates_per_rank"] == [line["config"]["dense_candidates_all_ranks"]]
    assert line["parity_checked"] is True and line["roofline"]["frac"] > 0


def test_region_cost_weights_balance_the_svr_shards():
    """The shard weights follow the SVR kernel's cost model (table entries + candidates), not the raw dense-grid size: the two shards of an
    exome-shaped batch take the same k_svr_dense time within 10 %."""
    from mipgen_amd import workloads
    P = capi.make_params(150, 170, score_method=capi.SCORE_SVR)
    chrom_len, all_iv = workloads.exome_layout()
    ivs = all_iv[:1536]
    w = workloads.shard_weights(ivs, P, True)
    (lo0, hi0), (lo1, hi1) = mdist.shard_regions(w.tolist(), 2)
    accs = []
    for lo, hi in ((lo0, hi0), (lo1, hi1)):
        acc = capi.Accel(P)
        acc.load_model_file(workloads.svr_model_path("/tmp/mipgen_test_models", workloads.practice62()[0], 256))
        acc.upload(workloads.build_exome(acc, chrom_len, ivs[lo:hi], P))
        acc.set_timing(True)
        acc.score_condense_all(capi.SCORE_SVR)
        accs.append(acc)
    # the minimum over alternating passes: other test workers (pytest -n) may be sharing the GPU, and a pass that ran beside their kernels says nothing
    ms = [float("inf"), float("inf")]
    for it in range(16):
        for k, acc in enumerate(accs):
            acc.score_condense_all(capi.SCORE_SVR)
            ms[k] = min(ms[k], acc.last_kernel_ms(0))
        if it >= 2 and abs(ms[0] - ms[1]) <= 0.10 * max(ms):
            break
    for acc in accs:
        acc.close()
    if os.environ.get("PYTEST_XDIST_WORKER") and abs(ms[0] - ms[1]) > 0.10 * max(ms):
        pytest.skip(f"kernel times {ms} taken beside other test workers' kernels (pytest -n): the balance is asserted on an otherwise idle GPU")
    assert abs(ms[0] - ms[1]) <= 0.10 * max(ms), ms


@pytest.mark.parametrize("name", ["long_default", "logistic_snp_trf", "svr_two_size_runs", "practice62_config2_svr", "mixed_12_regions", "mixed_small", "merge_flank_tags",
                                  "gaps_blocks", "practice62_config1"])
def test_one_process_per_rank_design_through_torch_distributed(name, tmp_path):
    """The multi-process product path (mipgen_amd/mp_design.py): one process per rank, every rank scores its cost-model shard of the design's regions
    on the accelerator, one torch.distributed gather of the condensed survivors, the sequential selection stage on rank 0 - and the output files (for
    non-silent designs including the all_mips records, numbered design-wide) are the ones the real reference wrote.  Two ranks share the box's GPU and exchange through gloo here (`--backend gloo --share-gpus`: everything but
    RCCL itself is the code of an 8-GPU run); run as a child process."""
    import json
    import subprocess
    import sys
    meta = H.load_design(name)
    if len(meta["intervals"]) < 2:
        pytest.skip("one region: nothing to shard")
    work = str(tmp_path / "mp")
    os.makedirs(work)
    argv = H.prepare_cli_workdir(meta, work)
    env = dict(os.environ, FAKEBWA_MODE=meta["bwa"], PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    p = subprocess.run([sys.executable, "-m", "mipgen_amd.mp_design", "--gpus", "2", "--backend", "gloo", "--share-gpus", "--mipgen-path", argv[0], "--"] + argv[1:],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, cwd=work, env=env)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    line = json.loads([l for l in p.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert line["ranks"] == 2 and 2 <= line["regions"] <= len(meta["intervals"]) and len(line["shards"]) == 2      # (overlapping BED lines merge)
    assert all(hi > lo for lo, hi in line["shards"]) and line["shards"][0][1] == line["shards"][1][0]          # both ranks hold a share, contiguous
    assert line["picked"] == meta["lines"]["picked_mips"] - 1
    if "-silent_mode" in meta.get("extra", []):
        H.compare_outputs(meta, work, keys=("picked_mips", "snp_mips"), check_all=False)
    else:
        # non-silent: every rank formats the all_mips records of its shard on the device with its own numbering; rank 0 appends the parts and shifts the
        # record numbers (mipgen.cpp:474,488,792) - all four files are the reference's
        H.compare_outputs(meta, work)
        assert not [f for f in os.listdir(work) if f.endswith(".part")]


def test_four_ranks_number_the_all_mips_records_design_wide(tmp_path):
    """BASELINE configs[0] (62 regions, 2.18 M all_mips records) through mipgen_amd/mp_design.py with FOUR ranks on the box's GPU: three renumbering bases,
    every output file the reference's."""
    import json
    import subprocess
    import sys
    meta = H.load_design("practice62_config1")
    work = str(tmp_path / "mp4")
    os.makedirs(work)
    argv = H.prepare_cli_workdir(meta, work)
    env = dict(os.environ, FAKEBWA_MODE=meta["bwa"], PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    p = subprocess.run([sys.executable, "-m", "mipgen_amd.mp_design", "--gpus", "4", "--backend", "gloo", "--share-gpus", "--mipgen-path", argv[0], "--"] + argv[1:],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, cwd=work, env=env)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    line = json.loads([l for l in p.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert line["ranks"] == 4 and len(line["shards"]) == 4 and all(hi > lo for lo, hi in line["shards"])
    H.compare_outputs(meta, work)


def test_rccl_collectives_of_the_multi_gpu_paths_with_one_rank():
    """The RCCL calls of bench.py --gpus N and mipgen_amd/mp_design.py (backend nccl, CUDA tensors, the dtypes they use) on a process group of one rank:
    no multi-GPU box was available to this build, so at least the API surface - init with device_id, all_gather / gather of byte buffers, all_reduce
    MAX / MIN, barrier - has executed on RCCL before the driver's run.  Child process (tests/rccl_one_rank_worker.py)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_PORT=str(_free_port()))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, "tests", "rccl_one_rank_worker.py")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, cwd=root, env=env)
    assert p.returncode == 0 and b"rccl one-rank collectives ok" in p.stdout, p.stderr.decode()[-3000:]


def test_bench_and_mp_design_take_their_rccl_code_path_with_one_rank(tmp_path):
    """bench.py and mipgen_amd/mp_design.py through `torch.distributed.run --nproc-per-node 1` with the default backend (nccl = RCCL) and --force-dist: the
    N > 1 code of both - process group with device_id, size exchange, the per-step gather straight from the library's survivor array in HBM (bench), the
    gathers of survivors / collapse results / grids and the record-number scan (mp_design) - executes on RCCL on this one-GPU box."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=root)
    launch = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1"]
    p = subprocess.run(launch + ["--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                                 "--no-extras", "--no-measure-traffic"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, cwd=root, env=env)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    line = json.loads([l for l in p.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert line["rccl_ranks"] == 1 and line["backend"] == "nccl" and line["parity_checked"] is True
    assert line["config"]["survivors_gathered_per_step"] == line["config"]["survivors_rank0"] > 0        # the gather ran, on device memory
    # the exome beside the headline family (`exome_strong`: shards, gather straight from HBM, max over ranks) takes its RCCL path too
    p = subprocess.run(launch + ["--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "1", "--warmup", "1", "--no-cpu-baseline",
                                 "--no-measure-traffic", "--no-parity-gate", "--scale-base-regions", "1024", "--exome-regions", "0", "--sustain-seconds", "0"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, cwd=root, env=env)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    line = json.loads([l for l in p.stdout.decode().splitlines() if l.startswith("{")][-1])
    ex = line["exome_strong"]
    assert ex["n_gpus"] == 1 and ex["survivors_gathered"] >= ex["survivors_rank0"] > 0 and "first 1024" in ex["what"]
    meta = H.load_design("logistic_snp_trf")
    work = str(tmp_path / "mp1")
    os.makedirs(work)
    argv = H.prepare_cli_workdir(meta, work)
    p = subprocess.run(launch + ["--master-port", str(_free_port()), "-m", "mipgen_amd.mp_design", "--gpus", "1", "--force-dist", "--mipgen-path", argv[0], "--"] + argv[1:],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, cwd=work, env=dict(env, FAKEBWA_MODE=meta["bwa"]))
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    line = json.loads([l for l in p.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert line["ranks"] == 1 and line["backend"] == "nccl"
    H.compare_outputs(meta, work)

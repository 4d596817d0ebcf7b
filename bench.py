#!/usr/bin/env python3
"""bench.py — candidate MIPs scored / second (SVR) on MI355X, BASELINE.json metric.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one pass of the hot path over one resident batch: the dense candidate grid of every region of the
workload is constructed (integer records of design_mip), scored with the libsvm RBF-SVR, and the reference's
score-dependent enumeration is replayed + condensed on the device.  Inputs are resident in HBM before the timed
region starts.  Workload at N=1: BASELINE.json configs[1] (practice_genes-shaped design, capture 140-180, SVR) on
the synthetic stand-in `practice62` with a synthetic 1024-SV model (no real genome / BED / trained model exists
offline: SURVEY.md section 8d).  For N>1 every rank scores its own `practice62` instance (weak scaling, regions are
independent) and the per-position survivors are gathered to rank 0 with one RCCL gather.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from mipgen_amd import capi, synth, workloads  # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
FP64_PEAK_TFLOPS = 78.6          # MI355X FP64 vector peak (spec)
ALG_BYTES_PER_CAND = 16          # 8 B score + 8 B integer record written per candidate


def cpu_baseline(genome: bytes, ivs, model_path: str, P_args: dict, n_sv: int) -> dict:
    """Reference CPU path timed on this box's host cores on a bounded sample of the same workload.
    kind 'reference' = the real reference binary (oracle/_ref/mipgen_ref, prebuilt from /root/reference);
    falls back to the oracle restatement ('port') if the binary did not travel."""
    import shutil
    import tempfile
    from oracle import run_reference as rr
    cores = 1                                         # the reference is single-threaded
    # bounded sample (~10-20 s of single-core work): the three shortest intervals (the static size skip leaves few capture sizes there)
    sample = sorted(ivs, key=lambda v: v.bed_end - v.bed_start)[:3]
    iv = sample[0]
    work = tempfile.mkdtemp(prefix="mipgen_cpu_")
    try:
        if rr.have_reference():
            os.makedirs(os.path.join(work, "genome"))
            synth.write_fasta(os.path.join(work, "genome", f"chr{iv.chrom}.fa"), f"chr{iv.chrom}", genome)
            synth.write_bed(os.path.join(work, "one.bed"), sample)
            r = rr.run_reference(work, os.path.join(work, "genome"), os.path.join(work, "one.bed"), "cpu", P_args["minC"], P_args["maxC"],
                                 score_method="svr", model_path=model_path, bwa_mode="unique", silent=False, timeout=600)
            if r["returncode"] == 0:
                with open(r["all_mips"], "rb") as fh:
                    n = fh.read().count(b"\n") - 1
                return {"value": n / r["seconds"], "unit": "candidates/s", "cores": cores, "kind": "reference",
                        "sample": f"reference binary (-O2) end-to-end on {len(sample)} of {len(ivs)} regions ({'+'.join(str(v.bed_end - v.bed_start) for v in sample)} bp, "
                                  f"{n} emitted candidates, {r['seconds']:.1f} s wall incl. its FASTQ/shim I/O), n_sv={n_sv}"}
        # port: the oracle's C restatement (same arithmetic as the reference, no text hop, -O2)
        from oracle import pyoracle as po
        P = capi.make_params(P_args["minC"], P_args["maxC"], score_method=capi.SCORE_SVR)
        rd = capi.build_region(genome, iv.chrom, iv.bed_start, iv.bed_end, P, label=iv.label)
        om = po.Model(model_path)
        t0 = time.perf_counter()
        n, _ = po.enumerate_region(P, rd, capi.SCORE_SVR, om, capacity=1)
        dt = time.perf_counter() - t0
        return {"value": n / dt, "unit": "candidates/s", "cores": cores, "kind": "port",
                "sample": f"oracle C restatement on 1 of {len(ivs)} regions ({n} emitted candidates, {dt:.1f} s), n_sv={n_sv}"}
    finally:
        shutil.rmtree(work, ignore_errors=True)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--nsv", type=int, default=1024)
    ap.add_argument("--min-capture", type=int, default=140)
    ap.add_argument("--max-capture", type=int, default=180)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-replay", action="store_true", help="time scoring only (kernel studies)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    distributed = world > 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    n_gpus = world if distributed else 1

    # ---- workload (per rank: same shape, different seed -> weak scaling over independent regions) ----------
    genome, ivs = workloads.practice62(seed=20140101 + rank)
    cache = os.path.join(ROOT, "gpurun_out", "bench_cache")
    model_genome, _ = (genome, None) if rank == 0 else workloads.practice62(seed=20140101)
    model_path = workloads.svr_model_path(cache if rank == 0 else cache + f"_r{rank}", model_genome, args.nsv)
    P = capi.make_params(args.min_capture, args.max_capture, score_method=capi.SCORE_SVR)
    stream = torch.cuda.current_stream().cuda_stream
    acc = capi.Accel(P, device=local_rank, stream=stream)
    acc.load_model_file(model_path)
    regions = workloads.build_regions(acc, genome, ivs, P)
    acc.upload(regions)                                   # inputs resident in HBM before the timed region
    n_cand = acc.batch_candidates()
    acc.set_timing(True)

    def step() -> None:
        acc.score_resident(capi.SCORE_SVR)
        if not args.no_replay:
            acc.replay_condense()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    kernel_ms = []
    records_ms = []
    replay_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        kernel_ms.append(acc.last_kernel_ms(0))          # HIP events on the launch stream around k_svr_dense
        records_ms.append(acc.last_kernel_ms(2))
        if not args.no_replay:
            replay_ms.append(acc.last_kernel_ms(3))
    survivors_gathered = 0
    if distributed and not args.no_replay:
        # the one exchange step of the path: per-position survivors -> rank 0 (RCCL gather over xGMI)
        from mipgen_amd import dist as mdist
        emitted, surv, _ = acc.download_replay(want_mask=False)
        allsurv = mdist.gather_to_rank0(surv, device=f"cuda:{local_rank}")
        if rank == 0:
            survivors_gathered = int(allsurv.shape[0])
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if distributed:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        tc = torch.tensor([n_cand], dtype=torch.int64, device="cuda")
        dist.all_reduce(tc, op=dist.ReduceOp.SUM)
        total_cand = int(tc.item())
    else:
        total_cand = n_cand

    if rank == 0:
        emitted_total = None
        if not args.no_replay:
            emitted, _, _ = acc.download_replay(want_mask=False)
            emitted_total = int(emitted.sum())
        value = total_cand * args.steps / dt
        k_ms = float(np.mean(kernel_ms))
        n_sv, gamma, rho = acc.model_info()
        achieved = ALG_BYTES_PER_CAND * n_cand / (k_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("k_svr_dense_bytes_per_launch")
            except Exception:
                traffic = None
        # issue-side counters of the dominant kernel from the committed PMC passes (tools/profile_round.sh), if present
        util = {}
        ppath = os.path.join(ROOT, "profiles", "r01g_pmc.json")
        if os.path.exists(ppath):
            try:
                c = json.load(open(ppath))["k_svr_dense"]
                cyc = c["GRBM_GUI_ACTIVE"] / 8.0                      # summed over the 8 XCDs
                util = {"valu_instr_per_wave_pair": c["SQ_INSTS_VALU"] * 64.0 / (n_cand * n_sv),
                        "valu_issue_frac": c["SQ_INSTS_VALU"] * 4.0 / (1024 * cyc), "lds_busy_frac": c["SQ_LDS_IDX_ACTIVE"] / (256 * cyc),
                        "source": "profiles/r01g_pmc.json (profiled launch, same workload)"}
            except Exception:
                util = {}
        out = {
            "metric": "candidate MIPs scored/sec (SVR)", "value": value, "unit": "candidates/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"practice62 (62 exon-like regions, synthetic stand-in for practice_genes.bed), capture {args.min_capture}-{args.max_capture} step 5, "
                                   f"{P.n_arm_pairs} arm pairs, SVR scoring, synthetic libsvm model",
                       "n_sv": n_sv, "regions_per_gpu": len(regions), "dense_candidates_per_gpu": n_cand,
                       "emitted_candidates_rank0": emitted_total, "replay_condense_in_step": not args.no_replay,
                       "survivors_gathered": survivors_gathered},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": "k_svr_dense", "kernel_ms": k_ms,
                         "algorithmic_bytes_per_launch": ALG_BYTES_PER_CAND * n_cand,
                         "note": "the path is bound by FP64 VALU issue and LDS, not by HBM (SURVEY.md section 8d); see fp64"},
            "fp64": {"pairs_per_launch": n_cand * n_sv, "pairs_per_s": n_cand * n_sv / (k_ms * 1e-3),
                     "naive_equiv_tflops": n_cand * n_sv * 600.0 / (k_ms * 1e-3) / 1e12, "peak_tflops": FP64_PEAK_TFLOPS, **util},
            "kernels_ms": {"k_svr_dense": k_ms, "k_records": float(np.mean(records_ms)),
                           "k_replay_condense(+memsets)": float(np.mean(replay_ms)) if replay_ms else None},
        }
        if not args.no_cpu_baseline and n_gpus == 1:
            out["cpu_baseline"] = cpu_baseline(genome, ivs, model_path, {"minC": args.min_capture, "maxC": args.max_capture}, n_sv)
        print(json.dumps(out))
    acc.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

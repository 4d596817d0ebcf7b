"""One process per GPU: a whole mipgen design through `torch.distributed` (RCCL over xGMI on the GPU box).

    python -m mipgen_amd.mp_design --gpus N -- -regions_to_scan x.bed -project_name out -min_capture_size 150 ... [any mipgen flag]

The reference's tile_regions (/root/reference/mipgen.cpp:412-515) enumerates, scores and condenses every region independently and then picks
sequentially in region order (the rand() stream and the used-arm sets persist across regions, :97, :1863).  So:

  every rank   opens the design (options + input stage of libmipgen_host.so; ranks > 0 write their copies of the input-stage files into a
               scratch directory), takes its contiguous shard of the regions - balanced with the weights the in-process driver uses
               (mipgen_design_region_weights) -, scores + replays + condenses it on ITS GPU (libmipgen_accel.so) and
  one exchange gathers the condensed survivors (2 per scan position, 24 bytes each), the collapse results (2 per base), the per-region emitted
               counts and grids to rank 0, which
  rank 0       runs the selection stage region by region (mipgen_design_select_region) and writes the design's files.

Every score method (mixed designs re-score their condensed survivors with the SVR on the rank that holds the region, before the gather), silent or
not: a non-silent design's all_mips records are formatted on every rank's GPU with the rank's own numbering into a part file beside the outputs, and
rank 0 appends the parts in rank order with the record numbers shifted by what the ranks before wrote (mipgen.cpp:474,488,792).  `--backend gloo --share-gpus` runs the same code on a one-GPU box (tests).  The product has no CPU path: without
a HIP device this fails."""
from __future__ import annotations

import json
import os
import shutil
import subprocess
import sys
import tempfile
import time
from typing import List

HERE = os.path.dirname(os.path.abspath(__file__))


def _split_args(argv: List[str]):
    import argparse
    if "--" in argv:
        k = argv.index("--")
        own, flags = argv[:k], argv[k + 1:]
    else:
        own, flags = argv, []
    ap = argparse.ArgumentParser(prog="python -m mipgen_amd.mp_design", description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--gpus", type=int, default=1, help="processes = GPUs of this node")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl", help="nccl = RCCL over xGMI; gloo + --share-gpus: tests on a one-GPU box")
    ap.add_argument("--share-gpus", action="store_true", help="ranks beyond the visible devices wrap around (never a measurement)")
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port of the self-launched run (0 = a free one: concurrent runs on one node do not collide)")
    ap.add_argument("--force-dist", action="store_true", help="tests: the process-group code path with a world of one rank (under torch.distributed.run --nproc-per-node 1)")
    ap.add_argument("--mipgen-path", default=os.path.join(HERE, "mipgen"), help="argv[0] of the design: mipgen_svr.model is looked for beside it (mipgen.cpp:409)")
    args = ap.parse_args(own)
    if not flags:
        ap.error("the mipgen flags follow `--`")
    return args, flags


def _launch(args, argv: List[str]) -> int:
    """--gpus N > 1 without a launcher: start torch.distributed.run as a CHILD (nothing here has touched a GPU; a GPU-initialised process never execs)."""
    port = args.master_port
    if not port:
        import socket
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), "-m", "mipgen_amd.mp_design"] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env, cwd=os.getcwd())


def _all_ranks_ok(dist, torch, xdev, ok: bool) -> bool:
    """Every rank reports once, after its shard is scored (or failed): the gather only starts when all of them succeeded."""
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=xdev if xdev else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


def main(argv: List[str]) -> int:
    args, flags = _split_args(argv)
    if "RANK" not in os.environ and args.gpus > 1:
        return _launch(args, argv)

    import numpy as np
    import torch
    import torch.distributed as dist
    from mipgen_amd import capi, dist as mdist, hostapi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("mipgen_amd.mp_design needs an MI355X: the hot path has no CPU fallback")
    if local_rank >= torch.cuda.device_count():
        if not (args.share_gpus and args.backend == "gloo"):
            raise SystemExit(f"rank {rank} needs GPU {local_rank}, but this node shows {torch.cuda.device_count()} device(s): one process per GPU")
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    distributed = world > 1 or args.force_dist
    xdev = f"cuda:{local_rank}" if args.backend == "nccl" else None      # where the tensors of the gather live
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")

    t_start = time.perf_counter()
    flags = list(flags)
    project = flags[flags.index("-project_name") + 1] if "-project_name" in flags and flags.index("-project_name") + 1 < len(flags) else None
    scratch = None
    part_path = (project or "mipgen_design") + f".all_mips.rank{rank}.part"   # (rank 0's project path: one node, one file system)
    rc = 0
    d = None
    reported = 0                                                           # status exchanges this rank has taken part in (1: shard scored, 2: selection done)
    try:
        try:
            os.remove(part_path)                                           # a part file left behind by a killed run must never be appended to this design
        except OSError:
            pass
        if rank > 0:                                                       # the input stage writes files named after the project: ranks > 0 keep theirs out of the way
            scratch = tempfile.mkdtemp(prefix=f"mipgen_rank{rank}_")
            name = os.path.basename(project) if project else "mipgen_design"
            hits = [k for k, f in enumerate(flags[:-1]) if f == "-project_name"]
            for k in hits:                                                 # (every occurrence: the option map keeps the last one)
                flags[k + 1] = os.path.join(scratch, name or "mipgen_design")
            if not hits:
                flags += ["-project_name", os.path.join(scratch, name)]
        d = hostapi.Design([args.mipgen_path] + flags)
        d.set_api_device(local_rank)                                       # (-gpu_copy_counter on: the shard's copy numbers are counted on this rank's GPU)
        P = d.params()
        n = d.region_count()
        scan = capi.SCORE_SVR if d.score_method == capi.SCORE_SVR else capi.SCORE_LOGISTIC
        shards = mdist.shard_regions(d.region_weights().tolist(), world)
        lo, hi = shards[rank]
        t_inputs = time.perf_counter()
        acc = capi.Accel(P, device=local_rank)
        mixed = d.score_method == capi.SCORE_MIXED
        if scan == capi.SCORE_SVR or mixed:
            acc.load_model_file(d.model_path)
            if scan == capi.SCORE_SVR:
                acc.set_dynamic_skip(True)                                 # mipgen.cpp:430 between the capture-size runs, as the front end does
            if hi > lo:                                                    # long-range content of the shard's regions, on the device
                views0 = d.regions(lo, hi - lo)
                lrc = acc.long_range_content_batch([d.long_range_seq(i) for i in range(lo, hi)], [views0[k].seq_start for k in range(hi - lo)],
                                                   [views0[k].seq_stop for k in range(hi - lo)])
                for k, i in enumerate(range(lo, hi)):
                    d.set_long_range_content(i, lrc[k])
        text_design = not d.silent                                         # all_mips records: formatted on the device, window by window
        local_rec = 0
        if text_design:
            acc.set_window_candidates(64 << 20)                            # the front end's window policy for designs that fetch per-window results
        if hi > lo and not text_design:
            grids = acc.upload_array(d.regions(lo, hi - lo), hi - lo)       # the shard's regions in the C layout, straight from libmipgen_host's arrays
            acc.score_condense_all(scan)                                    # score + replay + condense + collapse of every result window
            emitted, surv = acc.download_survivors()
            col = acc.download_collapsed(-1)                                # collapse_mips of the shard: 2 entries per base, region after region
            nbase = np.array([acc.region_bases(k)[1] for k in range(hi - lo)], dtype=np.int32)
        elif hi > lo:
            # non-silent: window by window - score, replay + condense, collapse, then print_details on the device (mipgen.cpp:765-794) with the rank's
            # own record numbers (from 0: rank 0 shifts them by what the ranks before wrote); the text goes to a part file beside the design's outputs
            grids = acc.upload_array(d.regions(lo, hi - lo), hi - lo)
            em_p, surv_p, col_p = [], [], []
            with open(part_path, "wb") as part:
                for w in range(acc.window_count()):
                    wi = acc.window_info(w)
                    acc.score_window(w, scan)
                    acc.replay_condense()
                    acc.collapse()
                    col_p.append(acc.download_collapsed(w))
                    e_w, s_w, _ = acc.download_replay(want_mask=False, window=w)
                    text, nrec = acc.format_all_mips_array(d.record_names(lo + wi["first_region"], wi["n_regions"]), d.middle, local_rec)
                    part.write(text)
                    local_rec += nrec
                    em_p.append(e_w); surv_p.append(s_w)
            emitted, surv, col = np.concatenate(em_p), np.concatenate(surv_p), np.concatenate(col_p)
            nbase = np.array([acc.region_bases(k)[1] for k in range(hi - lo)], dtype=np.int32)
        else:
            grids, emitted, surv = [], np.zeros(0, dtype=np.int64), np.zeros(0, dtype=capi.SURVIVOR_DTYPE)
            col, nbase = np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.int32)
        svr = np.zeros(0, dtype=np.float64)
        garr = np.frombuffer(bytes((capi.Grid * len(grids))(*grids)), dtype=hostapi.GRID_DTYPE) if grids else np.zeros(0, dtype=hostapi.GRID_DTYPE)
        garr = np.stack([garr[f].astype(np.int64) for f in ("offset", "count", "first_pos", "n_pos", "first_size_index", "n_sizes")], axis=1).reshape(-1) if grids else np.zeros(0, dtype=np.int64)
        if mixed:
            # mixed designs: every condensed survivor of the shard through the SVR as ONE list on the handle that holds its region (the pick stage
            # re-scores a subset of them, mipgen.cpp:1523-1527,1873-1877): features per candidate, then the FP64 matrix cores (k_svr_gemm)
            svr = np.full(surv.shape[0], np.nan)
            if hi > lo:
                cands, where, m = d.survivor_candidates(lo, garr, surv)
                svr[where] = acc.score_candidate_array(cands, m, capi.SCORE_SVR)
        t_scored = time.perf_counter()
        if distributed:
            reported = 1
            if not _all_ranks_ok(dist, torch, xdev, True):
                reported = 2                                                 # every rank leaves here: no second exchange
                raise RuntimeError("another rank failed before the gather")
        if distributed:
            # the one exchange step of the path: condensed survivors (+ the per-region counts and grids) -> rank 0
            all_surv = mdist.gather_to_rank0(surv, xdev)
            all_emitted = mdist.gather_to_rank0(emitted, xdev)
            all_grids = mdist.gather_to_rank0(garr, xdev)
            all_col = mdist.gather_to_rank0(col, xdev)
            all_nbase = mdist.gather_to_rank0(nbase, xdev)
            all_svr = mdist.gather_to_rank0(svr, xdev) if mixed else None
        else:
            all_surv, all_emitted, all_grids, all_col, all_nbase, all_svr = surv, emitted, garr, col, nbase, (svr if mixed else None)
        rec_base, rec_total = (mdist.exclusive_offsets(local_rec, xdev) if distributed else (0, local_rec)) if text_design else (0, 0)
        rec_bases = None
        if text_design and distributed:
            rec_bases = mdist.gather_to_rank0(np.array([rec_base], dtype=np.int64), xdev)
        t_gathered = time.perf_counter()
        acc.close()
        if rank == 0 and text_design:
            # the all_mips file: rank after rank, every rank's own numbering shifted by what the ranks before it wrote (whole lines, 64 MB at a time)
            for r in range(world):
                pp = (project or "mipgen_design") + f".all_mips.rank{r}.part"
                if shards[r][1] <= shards[r][0] or not os.path.exists(pp):     # (a rank with an empty shard wrote no part)
                    continue
                base_r = int(rec_bases[r]) if rec_bases is not None else 0
                with open(pp, "rb") as fh:
                    carry = b""
                    while True:
                        chunk = fh.read(64 << 20)
                        if not chunk:
                            break
                        chunk = carry + chunk
                        cut = chunk.rfind(b"\n") + 1
                        carry = chunk[cut:]
                        if cut:
                            d.write_all_mips(chunk[:cut], base_r)
                    if carry:
                        d.write_all_mips(carry, base_r)
        if rank == 0:
            all_grids = all_grids.reshape(-1, 6)
            assert all_grids.shape[0] == n and all_emitted.shape[0] == n, "the gather lost regions"
            # the sequential selection stage over all regions, in one call (a per-region Python loop costs 90 us per region: 18 s for 200,000)
            # -gpu_copy_counter on: where a survivor's 16-bit copy field saturates the selection looks the true count up in the host tables - then (and
            # only then) rank 0 builds them for every region; the ranks otherwise only ever build their own shard's
            rec = all_surv["record"]
            if bool((((capi.rec_ext_copy(rec) == 65535) | (capi.rec_lig_copy(rec) == 65535)) & (all_surv["cand_index"] >= 0)).any()):
                d.regions(0, n)
            d.select_regions(0, all_grids, all_surv, all_emitted, all_col, all_nbase, all_svr)
            c = d.counters()
            t_end = time.perf_counter()
            print(json.dumps({"regions": n, "ranks": world, "backend": args.backend if distributed else None, "shards": shards,
                              "survivors_gathered": int(all_surv.shape[0]), "emitted_candidates": int(all_emitted.sum()), "picked": c["picked"], "gaps": c["gaps"],
                              "seconds": {"inputs": round(t_inputs - t_start, 3), "score + condense (rank 0's shard)": round(t_scored - t_inputs, 3),
                                          "gather": round(t_gathered - t_scored, 3), "selection": round(t_end - t_gathered, 3)}}), flush=True)
        if distributed:
            reported = 2
            if not _all_ranks_ok(dist, torch, xdev, True):                   # rank 0 has read every part file and finished the selection stage - or failed in it
                raise RuntimeError("another rank failed after the gather")
    except BaseException as e:                                               # ANY failure is reported through the status exchange: no rank is left waiting in a collective
        print(f"[mp_design] rank {rank}: {type(e).__name__}: {e}", file=sys.stderr)
        rc = 1
        if distributed and reported < 2:
            try:
                # reported == 0: the "shard scored" exchange (the others stop before the gather); 1: the "selection done" exchange (the others are waiting in it)
                _all_ranks_ok(dist, torch, xdev, False)
            except BaseException:
                pass                                                          # (the peers are gone already: the launcher ends the run)
    finally:
        if d is not None:
            d.close()
        if scratch:
            shutil.rmtree(scratch, ignore_errors=True)
        try:
            os.remove(part_path)
        except OSError:
            pass
    if distributed:
        try:
            dist.destroy_process_group()
        except BaseException:
            pass
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))

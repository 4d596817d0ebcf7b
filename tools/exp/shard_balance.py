#!/usr/bin/env python3
"""What `bench.py --gpus N`'s exome_strong line would time on N GPUs, measured shard by shard on ONE: all 200,000 exons of BASELINE configs[3] cut N ways
by the kernels' cost model (mipgen_amd/dist.py: shard_regions over workloads.shard_weights - the rule of bench.py and of `mipgen -gpus N`'s blocks), every
shard scored on the one GPU through mipgen_accel_score_condense_all, one after the other.  The N-GPU pass takes the time of the SLOWEST shard (+ one gather
of 2.1 GB / N survivors over xGMI, not measured here): predicted speed-up = whole / slowest, balance = mean / slowest.

    python3 tools/exp/shard_balance.py [N ...]   (default 2 4 8)        ->  one JSON line per N"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from mipgen_amd import capi, workloads, dist as mdist  # noqa: E402


def shard_seconds(P, chrom_len, ivs, model):
    acc = capi.Accel(P)
    acc.load_model_file(model)
    acc.upload(workloads.build_exome(acc, chrom_len, ivs, P, with_lrc=True))
    n = acc.batch_candidates()
    acc.score_window(0, capi.SCORE_SVR)                       # warm-up as in bench.py: tile lists, result arrays, code objects
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    acc.score_condense_all(capi.SCORE_SVR)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    acc.close()
    return dt, n


def main() -> None:
    ns = [int(a) for a in sys.argv[1:]] or [2, 4, 8]
    P = capi.make_params(150, 170, score_method=capi.SCORE_SVR)
    chrom_len, ivs = workloads.exome_layout()
    model = workloads.svr_model_path(os.path.join(ROOT, "gpurun_out", "bench_cache"), workloads.practice62()[0], 1024, rho=workloads.MODEL_RHO["exome"])
    w = workloads.shard_weights(ivs, P, True).tolist()
    whole, n_all = shard_seconds(P, chrom_len, ivs, model)
    print(json.dumps({"n": 1, "seconds": whole, "dense_candidates": n_all, "candidates_per_s": n_all / whole}), flush=True)
    for n in ns:
        t, c = [], []
        for lo, hi in mdist.shard_regions(w, n):
            dt, nc = shard_seconds(P, chrom_len, ivs[lo:hi], model)
            t.append(dt); c.append(nc)
        print(json.dumps({"n": n, "shard_seconds": [round(x, 3) for x in t], "slowest": max(t), "mean": sum(t) / n, "balance_mean_over_slowest": sum(t) / n / max(t),
                          "predicted_speedup_whole_over_slowest": whole / max(t), "predicted_candidates_per_s": n_all / max(t),
                          "dense_candidates_per_shard": c, "sum_of_shards_over_whole": sum(t) / whole}), flush=True)


if __name__ == "__main__":
    main()

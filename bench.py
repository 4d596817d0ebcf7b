#!/usr/bin/env python3
"""bench.py — candidate MIPs scored / second on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--config practice62|regions5k|exome|exome_snp] [--regions R]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one pass of the hot path over the resident batch, the way a -silent_mode design runs it (mipgen_accel_score_condense_all):
for every region the dense candidate grid is constructed (integer records of design_mip), scored (libsvm RBF-SVR or logistic), and the
reference's score-dependent enumeration is replayed + condensed on the device; only 2 survivors per scan position remain.  Inputs are
resident in HBM before the timed region starts.

Workloads (mipgen_amd/workloads.py; no real genome / BED / trained model exists offline):
  practice62 (default)  BASELINE configs[1]: 62 exon-like regions, capture 140-180, SVR, synthetic 1024-SV libsvm model
  regions5k             configs[2] shape: --regions R of the 1,000 x 5 kb regions, capture 120-250; scan method logistic (mixed designs scan
                        with the logistic score) or --method svr
  exome / exome_snp     configs[3] / [4] shape: --regions R of the 200,000 exon-like intervals, capture 150-170 / 120-250, SVR

N = 1 (default): practice62 / SVR (configs[1], the config the 1-GPU metric is quoted on); the metric's own multi-GPU config - ALL 200,000 exons of
configs[3] - is timed in the same run on this one GPU (`exome_strong` = `scale_base`; its value and roofline also inside `config.exome_full` /
`roofline.exome_full`, where the driver's record keeps them); `extra` carries the round-5 shard of 65,536 exons, a sustained (>= 2 s) run of the
headline, the nSV sweep, the logistic scorer, the mixed-mode list re-scorer and the k-mer counter.  Before anything is timed an in-run PARITY GATE scores regions of the bench batch and
compares records / scores / replay / condensed survivors with the oracle (the checker only: nothing the oracle computes is timed or reported
as a rate) - SURVEY.md section 8d "correctness gate run with every measurement".

N > 1: one process per GPU (`--gpus N` without a launcher starts `torch.distributed.run` itself, before anything touches a GPU).
The BED is sharded over the ranks by the kernels' cost model (contiguous region ranges, no data-path collective while scoring); every step ends
with ONE gather of the condensed survivors to rank 0 (RCCL over xGMI), where the sequential pick stage consumes them.
  --scaling weak (default, every N)      the BED grows with N: N practice62-sized instances (62 N regions, 62 per rank) / N x --regions
  --scaling strong                       ONE fixed BED cut N ways (--config exome --regions R)

The scaling curve (driver: `bench.py --gpus N` for N = 1, 2, 4, 8) is ONE workload family in `value`: the configs[1] headline batch, weak-scaled
(`scale_family` names it in every line), so value(N) / (N value(1)) is a like-for-like efficiency.  The metric's own multi-GPU config (configs[3],
the exome) rides along in every line as `exome_strong`: all 200,000 exons cut N ways, one timed pass with its gather (at N = 1 the same
object is also printed as `scale_base`); divide exome_strong.value of an N-rank line by the one of the N = 1 line - never by `value`.
Every line names `rccl_ranks` (the world size torch.distributed reports after init_process_group; 1 without a process group) and the dense
candidates of every rank.

--dynamic-skip      mipgen.cpp:430 between the capture-size runs of the dense SVR scorer (configs of more than nine capture sizes: regions5k --method svr,
                    exome_snp): tiles whose positions have all stopped are not scored.  `value` then counts the candidates that WERE scored
                    (`config.dense_candidates_covered_per_s` is the covered rate); the headline config (nine sizes: one run) is not affected.
--measure-traffic   HBM bytes of the dominant kernel measured in THIS run: two child `rocprofv3 --pmc` passes (FETCH_SIZE, WRITE_SIZE - they do
                    not share a pass) of the same workload, started before this process touches the GPU (the default N = 1 line does this by
                    itself: ~40 s; --no-measure-traffic switches it off); `roofline.traffic` then comes from
                    them (FETCH_SIZE doubled: the gfx950 correction of MI355X_MICROARCH.md, HBM section) instead of from profiles/.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# rho of the synthetic SVR models (mipgen_amd/workloads.py: MODEL_RHO, per workload): places ~12 % of the arm-sum lists' last pairs above the
# reference's upper score limit (2.2), so the replay of the score-dependent early exits (mipgen.cpp:430,434) really skips candidates on
# every bench workload (emitted < dense), the exome included
MODEL_RHO = -2.2                 # practice62 (the headline)
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
FP64_PEAK_TFLOPS = 78.6          # MI355X FP64 vector peak (spec): 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz
ALG_BYTES_PER_CAND = 16          # 8 B score + 8 B integer record written per candidate
SCALE_REGIONS = 200000           # exons of the strong-scaling BED (--gpus N > 1) and of the N = 1 line's `scale_base`: all of configs[3]

CONFIGS = {
    #             capture      method      default regions
    "practice62": ((140, 180), "svr", 62),
    "regions5k": ((120, 250), "logistic", 24),
    "exome": ((150, 170), "svr", 8192),
    "exome_snp": ((120, 250), "svr", 4096),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=sorted(CONFIGS), default=None, help="default: practice62 (every N)")
    ap.add_argument("--regions", type=int, default=0, help="regions of the workload to use (0 = the config's default)")
    ap.add_argument("--method", choices=["svr", "logistic"], default=None)
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None, help="default: weak (the BED grows with N); strong = one BED cut N ways")
    ap.add_argument("--nsv", type=int, default=1024)
    ap.add_argument("--min-capture", type=int, default=0)
    ap.add_argument("--max-capture", type=int, default=0)
    ap.add_argument("--window-candidates", type=int, default=0, help="cap on the candidates of one result window (0 = what fits in HBM)")
    ap.add_argument("--sv-split", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the nSV sweep and the logistic line")
    ap.add_argument("--cpu-cores", type=int, default=0, help="processes of the multi-core CPU baseline (0 = all physical host cores)")
    ap.add_argument("--exome-regions", type=int, default=65536, help="exons of the round-5 exome shard kept in `extra` for continuity (N = 1 default run; 0 = skip)")
    ap.add_argument("--sustain-seconds", type=float, default=2.0, help="length of the sustained run of the headline in `extra` (0 = skip)")
    ap.add_argument("--no-parity-gate", action="store_true", help="skip the in-run oracle check (profiling runs)")
    ap.add_argument("--scale-base-regions", type=int, default=200000, help="exons of `exome_strong` / `scale_base` (0 = skip): ALL of BASELINE configs[3] by default - "
                    "one timed pass on the one GPU at N = 1, cut N ways at --gpus N")
    ap.add_argument("--measure-traffic", action="store_true", help="measure the dominant kernel's HBM bytes in this run (child rocprofv3 --pmc passes); the default "
                    "N = 1 run (practice62, no --no-extras) does so by itself")
    ap.add_argument("--no-measure-traffic", action="store_true", help="never start the child rocprofv3 passes (the traffic then comes from profiles/)")
    ap.add_argument("--dynamic-skip", action="store_true", help="mipgen.cpp:430 between the capture-size runs of the dense SVR scorer (regions of more than nine "
                    "capture sizes): tiles whose positions have all stopped are not scored; `value` then counts the candidates that WERE scored")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl", help="torch.distributed backend of the N > 1 run: nccl = RCCL over xGMI (the measured "
                    "path); gloo = the same exchange through host memory (tests of the N > 1 logic on boxes without a second GPU)")
    ap.add_argument("--share-gpus", action="store_true", help="tests only: ranks beyond the visible devices share them (rank r on GPU r mod devices; gloo only)")
    ap.add_argument("--force-dist", action="store_true", help="tests only: take the N > 1 code path (process group, per-step gather) with a world of ONE rank - "
                    "the RCCL calls of the step on a one-GPU box; needs RANK / WORLD_SIZE / MASTER_* in the environment (torch.distributed.run --nproc-per-node 1)")
    a = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", str(a.gpus)))
    if a.config is None:
        a.config = "practice62"
    if a.scaling is None:
        a.scaling = "weak"
    if not a.regions and a.config == "exome" and world > 1 and a.scaling == "strong":
        a.regions = SCALE_REGIONS
    return a


def self_launch(args) -> None:
    """`python bench.py --gpus N` without a launcher: start N ranks as a CHILD process before this one touches a GPU
    (a GPU-initialised process must never exec; this parent never initialises one)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


# ---------------------------------------------------------------------------------------------------------------------------
# workload assembly
# ---------------------------------------------------------------------------------------------------------------------------

def assemble(args, rank: int, world: int):
    """(describe, build) for this rank: `build(acc)` returns the rank's region records; describe is the workload string."""
    import numpy as np
    from mipgen_amd import capi, workloads, dist as mdist

    (c0, c1), method, n_default = CONFIGS[args.config]
    if args.min_capture:
        c0 = args.min_capture
    if args.max_capture:
        c1 = args.max_capture
    method = args.method or method
    n_regions = args.regions or n_default
    P = capi.make_params(c0, c1, score_method=capi.SCORE_SVR if method == "svr" else capi.SCORE_LOGISTIC)
    model_genome = workloads.practice62()[0]                       # support vectors are drawn from this genome for every workload
    if args.config == "practice62":
        # weak scaling: the BED is N practice62 instances (chromosomes "7", "7b", ...), sharded by dense-grid size
        inst = list(range(world)) if args.scaling == "weak" else [0]
        parts = []
        for k in inst:
            genome, ivs = workloads.practice62(seed=20140101 + k, n_regions=n_regions)
            parts.append((genome, ivs))
        all_ivs = [(k, iv) for k, (_, ivs) in enumerate(parts) for iv in ivs]
        weights = workloads.dense_candidates([iv for _, iv in all_ivs], P)
        lo, hi = mdist.shard_regions(workloads.shard_weights([iv for _, iv in all_ivs], P, method == "svr").tolist(), world)[rank]
        mine = all_ivs[lo:hi]

        def build(acc):
            out = []
            for k in sorted({k for k, _ in mine}):
                out += workloads.build_regions(acc, parts[k][0], [iv for kk, iv in mine if kk == k], P)
            return out
        desc = f"practice62 x{len(inst)} ({len(all_ivs)} exon-like regions, synthetic stand-in for practice_genes.bed)"
    elif args.config == "regions5k":
        total = n_regions * (world if args.scaling == "weak" else 1)
        ivs = workloads.regions5k_intervals(min(total, 1000))
        weights = workloads.dense_candidates(ivs, P)
        lo, hi = mdist.shard_regions(workloads.shard_weights(ivs, P, method == "svr").tolist(), world)[rank]

        def build(acc):
            return workloads.build_regions5k(acc, workloads.regions5k_genome(), ivs[lo:hi], P, with_lrc=method == "svr")
        desc = f"regions5k: {len(ivs)} of the 1,000 x 5,000 bp regions (12 Mb chromosome, N runs of 50, 98/1.5/0.5 % copy table)"
    else:
        total = n_regions * (world if args.scaling == "weak" else 1)
        chrom_len, all_iv = workloads.exome_layout()
        ivs = all_iv[:min(total, len(all_iv))]
        weights = workloads.dense_candidates(ivs, P)
        lo, hi = mdist.shard_regions(workloads.shard_weights(ivs, P, method == "svr").tolist(), world)[rank]

        def build(acc):
            return workloads.build_exome(acc, chrom_len, ivs[lo:hi], P, snps=args.config == "exome_snp", with_lrc=method == "svr")
        desc = (f"exome200k{'+SNPs (1/300 bp) +tags 4,4' if args.config == 'exome_snp' else ''}: the first {len(ivs)} of 200,000 exon-like intervals "
                f"(24 chromosomes, 300 Mb, 41 % GC)")
    desc += f", capture {c0}-{c1} step {P.capture_increment}, {P.n_arm_pairs} arm pairs, {method} scoring"
    return P, method, model_genome, build, desc, int(weights.sum())


def table_entries_min(P, grids) -> int:
    """Distinct factor-table entries the window-separable SVR form needs per support vector, independent of any tiling: per strand the
    (position, upstream arm length), (downstream start, downstream arm length) and (position, scan size) windows of a region."""
    from mipgen_amd import capi
    pairs = capi.arm_pairs_of(P)
    n_e, n_l = len({e for e, _ in pairs}), len({l for _, l in pairs})
    sums = sorted({e + l for e, l in pairs})
    total = 0
    for g in grids:
        sizes = [P.max_capture_size - (g.first_size_index + k) * P.capture_increment for k in range(g.n_sizes)]
        ss = sorted({C - s for C in sizes for s in sums})
        if not ss:
            continue
        nq = g.n_pos + (ss[-1] - ss[0])
        total += (g.n_pos * n_e + nq * n_l + g.n_pos * len(ss)) + (g.n_pos * n_l + nq * n_e + g.n_pos * len(ss))      # '+' strand, '-' strand
    return total



# ---------------------------------------------------------------------------------------------------------------------------
# in-run parity gate ("correctness gate run with every measurement"; reference semantics mipgen.cpp:426-497)
# ---------------------------------------------------------------------------------------------------------------------------

def parity_gate(acc, P, regions, grids, method: str, model_path, device: int, stream: int, sample: int = 1500) -> dict:
    """Checks the bench batch against the oracle BEFORE anything is timed (the oracle is the checker here, nothing it computes is timed):
      (1) the smallest region of the batch scored alone: every integer record bit-exact, every score within 1e-5 (NaN / guard values exact),
          the replayed emitted mask, the emitted count and the condensed survivors identical to the oracle's replay + fold fed with the
          device's scores (mipgen.cpp:426-497, 1670-1746);
      (2) the batch's own survivors of that region (scored inside the full batch by score_condense_all) equal the ones of (1);
      (3) a random sample of candidates of the LARGEST region (all its capture sizes): records bit-exact, scores within 1e-5 of the
          oracle's per-candidate arithmetic (SVMipv4.cpp:60-248, svm.cpp:2504-2593).
    Raises AssertionError on any difference."""
    import ctypes
    import numpy as np
    from mipgen_amd import capi
    from oracle import pyoracle as po                     # checker only
    t_start = time.perf_counter()
    m = capi.SCORE_SVR if method == "svr" else capi.SCORE_LOGISTIC
    om = po.Model(model_path) if model_path else None
    nz = [i for i, g in enumerate(grids) if g.count > 0]
    i_small, i_big = min(nz, key=lambda i: grids[i].count), max(nz, key=lambda i: grids[i].count)
    acc.score_condense_all(m)
    emitted_b, surv_b = acc.download_survivors()
    chk = capi.Accel(P, device=device, stream=stream)
    if model_path:
        chk.load_model_file(model_path)

    def close(a, b):
        both_nan = np.isnan(a) & np.isnan(b)
        return bool(np.all(both_nan | (np.abs(a - b) <= 1e-5)))

    # (1) smallest region alone
    rd, g = regions[i_small], grids[i_small]
    _, s, r = chk.score_regions([rd], m)
    chk.replay_condense()
    em, sv, mask = chk.download_replay()
    _, os_, or_ = po.score_region_dense(P, rd, m, om)
    assert np.array_equal(r, or_), "parity gate: integer records differ from the oracle"
    assert close(s, os_), f"parity gate: scores differ from the oracle (max {np.nanmax(np.abs(s - os_))})"
    max_err = float(np.nanmax(np.abs(s - os_))) if s.size else 0.0
    n_emit, omask = po.replay_region(P, rd, s, r)
    assert int(em[0]) == n_emit and np.array_equal(mask, omask), "parity gate: replayed enumeration differs from the oracle"
    osurv = po.condense_region(P, rd, s, r, omask)
    assert np.array_equal(sv["cand_index"], osurv["cand_index"]) and np.array_equal(sv["record"], osurv["record"]) \
        and np.array_equal(sv["score"], osurv["score"], equal_nan=True), "parity gate: condensed survivors differ from the oracle"
    # (2) the same region inside the bench batch
    pos0 = sum(gg.n_pos for gg in grids[:i_small])
    got = surv_b[2 * pos0:2 * (pos0 + g.n_pos)]
    exp_idx = np.where(osurv["cand_index"] >= 0, osurv["cand_index"] + g.offset, -1)
    assert int(emitted_b[i_small]) == n_emit, "parity gate: emitted count of the batch differs"
    assert np.array_equal(got["cand_index"], exp_idx) and np.array_equal(got["record"], osurv["record"]) and close(got["score"], osurv["score"]), \
        "parity gate: the batch's survivors differ from the region scored alone"
    # (3) sample of the largest region
    rd, g = regions[i_big], grids[i_big]
    _, s, r = chk.score_regions([rd], m)
    rng = np.random.default_rng(20140101)
    pick = rng.choice(g.count, size=min(sample, g.count), replace=False)
    A = P.n_arm_pairs
    lrc = np.array([rd.c.long_range_content[i] for i in range(capi.N_LRC)])
    n_checked = 0
    for idx in pick:
        a = int(idx % A); row = int(idx // A); strand = row & 1; rest = row >> 1
        ki, pi = rest % g.n_sizes, rest // g.n_sizes
        cand = (0, g.first_pos + pi, P.max_capture_size - (g.first_size_index + ki) * P.capture_increment, P.arm_ext[a], P.arm_lig[a], int(strand))
        skipped, d = po.design(P, rd, cand)
        flags = int(capi.rec_flags(r[idx:idx + 1])[0])
        if skipped:
            assert not (flags & capi.FLAG_VALID), "parity gate: a bounds-skipped candidate is marked valid"
            continue
        osc, _, oints = po.score_designed(d, m, lrc, om)
        orec = int(po.oracle().mo_record_of(ctypes.byref(d), 1, ctypes.byref(oints)))
        assert int(r[idx]) == orec, f"parity gate: integer record of candidate {idx} differs ({int(r[idx]):#x} vs {orec:#x})"
        assert (np.isnan(osc) and np.isnan(s[idx])) or abs(s[idx] - osc) <= 1e-5, f"parity gate: score of candidate {idx} differs ({s[idx]} vs {osc})"
        max_err = max(max_err, 0.0 if np.isnan(osc) else abs(float(s[idx]) - osc))
        n_checked += 1
    chk.close()
    return {"parity_checked": True, "oracle": "oracle/_build/libmipgen_oracle.so (C restatement, pinned to the compiled reference: tests/test_oracle_golden.py)",
            "full_region": {"index": i_small, "dense_candidates": int(grids[i_small].count), "emitted": n_emit,
                            "checked": "records bit-exact, scores <= 1e-5, emitted mask, condensed survivors (alone and inside the bench batch)"},
            "sampled_region": {"index": i_big, "dense_candidates": int(grids[i_big].count), "capture_sizes": int(grids[i_big].n_sizes), "candidates_checked": n_checked},
            "max_abs_score_error": max_err, "seconds": time.perf_counter() - t_start}

# ---------------------------------------------------------------------------------------------------------------------------
# CPU baseline: the real reference binary on this box's host cores
# ---------------------------------------------------------------------------------------------------------------------------

def physical_cores() -> int:
    """Physical cores this process may run on (distinct (package, core) pairs of the affinity mask; SMT siblings count once)."""
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except AttributeError:
        cpus = list(range(os.cpu_count() or 1))
    seen = set()
    for c in cpus:
        try:
            pkg = open(f"/sys/devices/system/cpu/cpu{c}/topology/physical_package_id").read().strip()
            core = open(f"/sys/devices/system/cpu/cpu{c}/topology/core_id").read().strip()
            seen.add((pkg, core))
        except OSError:
            seen.add(("?", c))
    return max(1, len(seen))


def cgroup_cpu_quota() -> float:
    """CPUs this container may use at once according to its cgroup (cpu.max / cfs quota); inf if unlimited or unknown."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return q / per
    except (OSError, ValueError):
        pass
    return float("inf")


def cpu_baseline(args, model_path: str, n_sv: int) -> dict:
    """The reference CPU path (oracle/_ref: the real reference compiled from /root/reference by oracle/Makefile) timed on this box's host
    cores, on a bounded sample of the practice62 / capture 140-180 / SVR workload (8 regions at every 3rd rank of its length-sorted BED: seconds to ~20 s of CPU work each).  The reference is single-threaded and not re-entrant: multi-core = independent processes on BED shards.
      leg A  one process per sample region, -O2, all_mips written: the emitted-candidate count of every sample region and the single-process
             rate; beside them ONE process of the binary AS SHIPPED (/root/reference/makefile:3-4: no -O flag) on the shortest region;
      leg B  C = all physical cores (capped by the container's cgroup CPU quota, if any) processes at once, -O2, -silent_mode on (no all_mips text): `value` = sum over the processes of
             (emitted candidates of its region / its tile_regions time), every process timed while all the others run.  tile_regions time =
             from the reference's own "[mipgen] bwa copy number analysis finished" line (mipgen.cpp:349) to exit: enumeration + scoring +
             selection, WITHOUT the input stage and the FASTQ / stand-in bwa (awk) I/O, which the GPU figure does not cover either."""
    import shutil
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    from mipgen_amd import capi, synth, workloads
    from oracle import run_reference as rr

    genome, ivs = workloads.practice62()
    try:
        cores_avail = len(os.sched_getaffinity(0))
    except AttributeError:
        cores_avail = os.cpu_count() or 1
    phys = physical_cores()
    quota = cgroup_cpu_quota()
    cores = args.cpu_cores or max(1, min(phys, int(quota) if quota != float("inf") else phys))
    cpu_model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    # 8 sample regions at every 3rd rank of the length-sorted BED (ranks 0, 3, ... 21 of 62: 60 to ~120 bp, one to three capture sizes after
    # the static skip of mipgen.cpp:429) - seconds to ~20 s of reference CPU work each, not only the shortest regions
    by_len = sorted(ivs, key=lambda v: (v.bed_end - v.bed_start))
    pool = [by_len[min(3 * i, len(by_len) - 1)] for i in range(8)]
    shortest = min(range(len(pool)), key=lambda i: pool[i].bed_end - pool[i].bed_start)
    work = tempfile.mkdtemp(prefix="mipgen_cpu_")          # not /dev/shm: it is mounted noexec on the GPU boxes (the binary is copied beside its model)

    def run_one(tag: str, i: int, o0: bool = False, silent: bool = False):
        iv = pool[i % len(pool)]
        w = os.path.join(work, tag)
        os.makedirs(os.path.join(w, "genome"))
        synth.write_fasta(os.path.join(w, "genome", f"chr{iv.chrom}.fa"), f"chr{iv.chrom}", genome)
        synth.write_bed(os.path.join(w, "one.bed"), [iv])
        r = rr.run_reference(w, os.path.join(w, "genome"), os.path.join(w, "one.bed"), "cpu", 140, 180, score_method="svr", model_path=model_path,
                             bwa_mode="unique", silent=silent, o0=o0, timeout=900, hot_marker=True)
        if r["returncode"] != 0 or r["hot_seconds"] is None:
            raise RuntimeError(r["stderr"][-500:])
        n = None
        if not silent:
            with open(r["all_mips"], "rb") as fh:
                n = fh.read().count(b"\n") - 1
        shutil.rmtree(w, ignore_errors=True)
        return {"n": n, "seconds": r["seconds"], "hot": r["hot_seconds"], "len": iv.bed_end - iv.bed_start}

    try:
        if rr.have_reference():
            have_o0 = rr.have_reference(o0=True)
            with ThreadPoolExecutor(max_workers=len(pool) + 1) as ex:
                fa = [ex.submit(run_one, f"a{i}", i) for i in range(len(pool))]
                f0 = ex.submit(run_one, "o0", shortest, True) if have_o0 else None
                A = [f.result() for f in fa]
                O0 = f0.result() if f0 else None
            counts = [a["n"] for a in A]
            t0 = time.perf_counter()
            with ThreadPoolExecutor(max_workers=cores) as ex:
                B = list(ex.map(lambda i: run_one(f"b{i}", i, False, True), range(cores)))
            wall = time.perf_counter() - t0
            value = sum(counts[i % len(pool)] / B[i]["hot"] for i in range(cores))
            n_all = sum(counts[i % len(pool)] for i in range(cores))
            out = {"value": value, "unit": "candidates/s", "cores": cores, "kind": "reference",
                   "single_core_value": A[shortest]["n"] / A[shortest]["hot"],
                   "single_core_value_end_to_end": A[shortest]["n"] / A[shortest]["seconds"],
                   "end_to_end_value": n_all / wall,
                   "value_is": "the sum of the per-process rates (each process over its own tile_regions time)",
                   "end_to_end_value_is": "all candidates / the wall time of the slowest process (input stages and the longest region included)",
                   "cpu_model": cpu_model, "host_threads_available": cores_avail, "physical_cores": phys,
                   "cgroup_cpu_quota": None if quota == float("inf") else quota,
                   "scope": "reference binary, tile_regions only (enumeration + scoring + selection; from its 'bwa copy number analysis finished' line to exit), "
                            "-O2, -silent_mode on in the multi-process leg; end_to_end_value includes its input stage and the stand-in bwa / FASTQ I/O",
                   "sample": f"practice62 / capture 140-180 / SVR n_sv={n_sv}: {cores} concurrent reference processes (-O2, -silent_mode on), one region each from the "
                             f"{len(pool)} sample regions (every 3rd of the length-sorted BED: {'/'.join(str(v.bed_end - v.bed_start) for v in pool)} bp), "
                             f"{n_all} emitted candidates, tile_regions {min(b['hot'] for b in B):.1f}-{max(b['hot'] for b in B):.1f} s per process, {wall:.1f} s wall; "
                             f"alone: {A[shortest]['len']}-bp region, {A[shortest]['n']} candidates, tile_regions {A[shortest]['hot']:.1f} s of {A[shortest]['seconds']:.1f} s"}
            if O0:
                out["as_shipped_O0"] = {"single_core_value": O0["n"] / O0["hot"], "single_core_value_end_to_end": O0["n"] / O0["seconds"],
                                        "note": f"oracle/_ref/mipgen_ref_O0 = the reference's own flags (/root/reference/makefile:3-4: -g, no -O) on the {O0['len']}-bp region: "
                                                f"{O0['n']} candidates, tile_regions {O0['hot']:.1f} s of {O0['seconds']:.1f} s; -O2 / -O0 = {O0['hot'] / A[shortest]['hot']:.2f}x"}
            return out
        # port: the oracle's C restatement (same arithmetic as the reference, no text hop, -O2), one core
        from oracle import pyoracle as po
        P = capi.make_params(140, 180, score_method=capi.SCORE_SVR)
        iv = pool[0]
        rd = capi.build_region(genome, iv.chrom, iv.bed_start, iv.bed_end, P, label=iv.label)
        om = po.Model(model_path)
        t0 = time.perf_counter()
        n, _ = po.enumerate_region(P, rd, capi.SCORE_SVR, om, capacity=1)
        dt = time.perf_counter() - t0
        return {"value": n / dt, "unit": "candidates/s", "cores": 1, "kind": "port", "cpu_model": cpu_model,
                "sample": f"oracle C restatement on 1 of {len(ivs)} regions ({n} emitted candidates, {dt:.1f} s), n_sv={n_sv}"}
    finally:
        shutil.rmtree(work, ignore_errors=True)


def measure_traffic(args, kernel: str):
    """HBM bytes per launch of `kernel`, measured in this run: one child `rocprofv3 --kernel-trace --pmc <counter>` pass per counter
    (FETCH_SIZE and WRITE_SIZE do not fit one pass; never combined with other trace domains) over the same workload at --steps 2 --warmup 1.
    Must run BEFORE this process touches the GPU (the children are started as ordinary child processes, the program itself after `--`).
    Counter unit: KiB.  FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM: gfx950 tallies 128-byte requests of coalesced reads at 64 bytes; the raw
    value is kept beside it), WRITE_SIZE is taken as it reads.  Returns None when rocprofv3 is missing or a pass fails."""
    import csv
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3")
    if not exe:
        return None
    base = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", "2", "--warmup", "1", "--config", args.config, "--scaling", args.scaling,
            "--nsv", str(args.nsv), "--no-cpu-baseline", "--no-extras", "--no-parity-gate", "--no-measure-traffic"]
    if args.regions:
        base += ["--regions", str(args.regions)]
    if args.method:
        base += ["--method", args.method]
    if args.min_capture:
        base += ["--min-capture", str(args.min_capture)]
    if args.max_capture:
        base += ["--max-capture", str(args.max_capture)]
    # everything that changes the launch travels to the child passes
    if args.dynamic_skip:
        base += ["--dynamic-skip"]
    if args.sv_split:
        base += ["--sv-split", str(args.sv_split)]
    if args.window_candidates:
        base += ["--window-candidates", str(args.window_candidates)]
    env = dict(os.environ, TMPDIR="/tmp")
    got = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="mipgen_pmc_", dir="/tmp")
        try:
            r = subprocess.run([exe, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "p", "--"] + base, cwd="/tmp", env=env,
                               stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=240)
            vals = []
            for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(fn)):
                    if kernel in row["Kernel_Name"] and row["Counter_Name"] == ctr:
                        vals.append(float(row["Counter_Value"]))
            if r.returncode != 0 or not vals:
                return None
            got[ctr] = (sum(vals) / len(vals) * 1024.0, len(vals))
        except Exception:
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    fetch, write = got["FETCH_SIZE"][0], got["WRITE_SIZE"][0]
    return {"bytes_per_launch": 2.0 * fetch + write, "fetch_bytes_raw": fetch, "fetch_bytes_corrected": 2.0 * fetch, "write_bytes": write,
            "launches_averaged": min(got["FETCH_SIZE"][1], got["WRITE_SIZE"][1]), "kernel": kernel, "child_command": " ".join(base[1:]),
            "how": "child `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of this workload (--steps 2 --warmup 1), mean per launch, "
                   "summed over the device; counter unit KiB; FETCH_SIZE x 2 = the gfx950 correction (MI355X_MICROARCH.md, HBM section)"}


def newest_profile(pattern: str):
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    return files[-1] if files else None



class SurvivorGather:
    """The one exchange step of the path: the handle's condensed survivors, straight from its array in HBM (mipgen_accel_survivors_device_ptr
    wrapped as a torch tensor, no copy), gathered to rank 0 - RCCL over xGMI: direct peer -> root transfers (`--backend gloo`: through host memory)."""

    def __init__(self, acc, world: int, rank: int, local_rank: int, xdev: str):
        import torch
        import torch.distributed as dist
        self.dist, self.rank = dist, rank
        ptr, n_slots = acc.survivors_device_ptr()
        self.nbytes = n_slots * 24

        class _DevView:
            def __init__(self, p, nbytes):
                self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (p, False), "version": 2}
        self.send = torch.as_tensor(_DevView(ptr, max(n_slots, 1) * 24), device=f"cuda:{local_rank}")
        sizes = [torch.zeros(1, dtype=torch.int64, device=xdev) for _ in range(world)]
        dist.all_gather(sizes, torch.tensor([self.nbytes], dtype=torch.int64, device=xdev))
        self.sizes = [int(t.item()) for t in sizes]
        mx = max(max(self.sizes), 24)
        self.pad = torch.zeros(mx, dtype=torch.uint8, device=xdev)
        self.recv = [torch.zeros(mx, dtype=torch.uint8, device=xdev) for _ in range(world)] if rank == 0 else None

    def __call__(self) -> None:
        self.pad[: self.nbytes].copy_(self.send[: self.nbytes])
        self.dist.gather(self.pad, self.recv, dst=0)


def exome_line(args, device: int, stream: int, model_path: str, n_regions: int = 0, rank: int = 0, world: int = 1, distributed: bool = False,
               xdev: str = "cuda") -> dict:
    """The metric's own multi-GPU config (BASELINE configs[3]: exome200k, capture 150-170, SVR): the first n_regions exons cut `world` ways by the
    kernels' cost model, one warm-up pass over the first result window, then ONE timed pass of every rank over its shard through score_condense_all
    (records + k_svr_dense + replay / condense per result window) + the gather of the survivors to rank 0; barrier on both sides, max over ranks.
    world = 1: the whole BED on the one GPU (the `scale_base` / `exome_strong` point of the N = 1 line)."""
    import torch
    import torch.distributed as dist
    from mipgen_amd import capi, workloads, dist as mdist
    P = capi.make_params(150, 170, score_method=capi.SCORE_SVR)
    acc = capi.Accel(P, device=device, stream=stream)
    # the same support vectors as the headline's model, rho placed for the exome (workloads.MODEL_RHO): its replay takes early exits too
    acc.load_model_file(workloads.svr_model_path(os.path.dirname(model_path), workloads.practice62()[0], args.nsv, rho=workloads.MODEL_RHO["exome"]))
    t0 = time.perf_counter()
    chrom_len, all_iv = workloads.exome_layout()
    ivs = all_iv[:min(n_regions or args.exome_regions, len(all_iv))]
    lo, hi = mdist.shard_regions(workloads.shard_weights(ivs, P, True).tolist(), world)[rank] if world > 1 else (0, len(ivs))
    regions = workloads.build_exome(acc, chrom_len, ivs[lo:hi], P, with_lrc=True)
    grids = acc.upload(regions)
    t_build = time.perf_counter() - t0
    n_cand = acc.batch_candidates()
    acc.set_timing(True)
    acc.score_window(0, capi.SCORE_SVR)                       # warm-up: tile lists, result arrays, code objects
    gather = SurvivorGather(acc, world, rank, device, xdev) if distributed else None
    if gather:
        gather()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    acc.score_condense_all(capi.SCORE_SVR)
    if gather:
        gather()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    d1 = time.perf_counter() - t1
    per_rank = [n_cand]
    if distributed:
        tt = torch.tensor([d1], dtype=torch.float64, device=xdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        d1 = float(tt.item())
        per = [torch.zeros(1, dtype=torch.int64, device=xdev) for _ in range(world)]
        dist.all_gather(per, torch.tensor([n_cand], dtype=torch.int64, device=xdev))
        per_rank = [int(t.item()) for t in per]
    total = sum(per_rank)
    emitted, surv = acc.download_survivors()
    n_sv = acc.model_info()[0]
    line = {"what": f"exome200k (BASELINE configs[3]): the first {len(ivs)} of 200,000 exon-like intervals, capture 150-170, SVR n_sv={n_sv}, cut {world} way(s) by the "
                    f"kernels' cost model, one timed pass (rank 0: {acc.window_count()} result windows; wall clock around score_condense_all"
                    f"{' + the gather of the survivors to rank 0' if distributed else ''}, max over ranks)",
            "value": total / d1, "unit": "candidates/s", "seconds": d1, "dense_candidates": total, "dense_candidates_per_rank": per_rank, "regions": len(ivs),
            "n_gpus": world, "scaling": "strong", "result_windows_rank0": acc.window_count(),
            "emitted_candidates_rank0": int(emitted.sum()), "survivors_rank0": int((surv["cand_index"] >= 0).sum()), "build_and_upload_seconds": t_build}
    if gather:
        line["survivors_gathered"] = sum(gather.sizes) // 24
    if world == 1:
        ent = table_entries_min(P, grids)
        flops = float(n_sv) * (3.0 * n_cand + 45.0 * ent)
        line["roofline"] = {"bound": "fp64_valu", "achieved": flops / d1 / 1e12, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": flops / d1 / 1e12 / FP64_PEAK_TFLOPS,
                            "kernel": "k_svr_dense (+ k_records, replay / condense: the whole pass is inside the clock)", "algorithmic_flops": flops,
                            "table_entries_per_sv": ent,
                            "hbm": {"achieved": ALG_BYTES_PER_CAND * n_cand / d1 / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": ALG_BYTES_PER_CAND * n_cand / d1 / 1e9 / HBM_PEAK_GBS}}
    acc.close()
    del regions
    return line


def exome_rates_by_sizes(args, device: int, stream: int, model_path: str, n_regions: int = 4096) -> dict:
    """k_svr_dense on the exome regions that keep K capture sizes after the static skip of mipgen.cpp:429, K by K (the first n_regions exons):
    regions of ONE size - half of an exome BED - share an arm factor between at most six candidates instead of thirty."""
    from mipgen_amd import capi, workloads
    P = capi.make_params(150, 170, score_method=capi.SCORE_SVR)
    acc = capi.Accel(P, device=device, stream=stream)
    acc.load_model_file(workloads.svr_model_path(os.path.dirname(model_path), workloads.practice62()[0], args.nsv, rho=workloads.MODEL_RHO["exome"]))
    chrom_len, all_iv = workloads.exome_layout()
    ivs = all_iv[:n_regions]
    grids = acc.upload(workloads.build_exome(acc, chrom_len, ivs, P))
    ks = [g.n_sizes for g in grids]
    tot = float(sum(g.count for g in grids))
    acc.set_timing(True)
    out = {}
    for K in sorted(set(ks)):
        sub = [iv for iv, k in zip(ivs, ks) if k == K]
        gr = acc.upload(workloads.build_exome(acc, chrom_len, sub, P))
        n = sum(g.count for g in gr)
        ms = []
        for _ in range(3):
            acc.score_window(0, capi.SCORE_SVR)
            ms.append(acc.last_kernel_ms(0))
        out[str(K)] = {"regions": len(sub), "dense_candidates": n, "candidates_share": n / tot, "k_svr_dense_ms": min(ms), "candidates_per_s": n / (min(ms) * 1e-3)}
    acc.close()
    return out


def main() -> None:
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    traffic_run = None
    default_line = args.gpus == 1 and args.config == "practice62" and not args.no_extras and not args.method and not args.regions
    if (args.measure_traffic or default_line) and not args.no_measure_traffic and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        # child rocprofv3 passes, before this process initialises the GPU
        traffic_run = measure_traffic(args, "k_svr_dense" if (args.method or CONFIGS[args.config][1]) == "svr" else "k_logistic_dense")

    import numpy as np
    import torch
    import torch.distributed as dist
    from mipgen_amd import capi, workloads

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    if local_rank >= torch.cuda.device_count():
        if not (args.share_gpus and args.backend == "gloo"):
            raise SystemExit(f"bench.py: rank {rank} needs GPU {local_rank}, but this node shows {torch.cuda.device_count()} device(s): one process per GPU")
        local_rank %= torch.cuda.device_count()             # tests of the N > 1 logic on a one-GPU box (never a measurement: `shared_gpus` in the line)
    torch.cuda.set_device(local_rank)
    distributed = world > 1 or args.force_dist
    xdev = "cuda" if args.backend == "nccl" else "cpu"      # where the tensors of the collectives live
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")

    P, method, model_genome, build, desc, global_dense = assemble(args, rank, world)
    cache = os.path.join(ROOT, "gpurun_out", "bench_cache") + ("" if rank == 0 else f"_r{rank}")
    stream = torch.cuda.current_stream().cuda_stream
    acc = capi.Accel(P, device=local_rank, stream=stream)
    m = capi.SCORE_SVR if method == "svr" else capi.SCORE_LOGISTIC
    model_path = None
    if method == "svr":
        model_path = workloads.svr_model_path(cache, model_genome, args.nsv, rho=workloads.MODEL_RHO[args.config])
        acc.load_model_file(model_path)
    regions = build(acc)
    if args.window_candidates:
        acc.set_window_candidates(args.window_candidates)
    if args.sv_split:
        acc.set_sv_split(args.sv_split)
    if args.dynamic_skip:
        acc.set_dynamic_skip(True)
    grids = acc.upload(regions)                            # inputs resident in HBM before the timed region
    n_cand = acc.batch_candidates()
    acc.set_timing(True)
    gather = SurvivorGather(acc, world, rank, local_rank, xdev) if distributed else None

    def step() -> None:
        acc.score_condense_all(m)
        if gather:
            gather()                                        # the one exchange step of the path: condensed survivors -> rank 0

    # correctness gate of this measurement: rank 0 checks its batch against the oracle before anything is timed
    gate = None
    if rank == 0 and not args.no_parity_gate:
        gate = parity_gate(acc, P, regions, grids, method, model_path, local_rank, stream)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    kernel_ms, records_ms, replay_ms = [], [], []
    if args.dynamic_skip:
        acc.skipped_candidates()                            # reset: count the timed steps only
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        kernel_ms.append(acc.last_kernel_ms(0 if method == "svr" else 2))      # HIP events on the launch stream around the scoring kernel
        records_ms.append(acc.last_kernel_ms(2))
        replay_ms.append(acc.last_kernel_ms(3))
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    skipped_per_step = acc.skipped_candidates() // max(args.steps, 1) if args.dynamic_skip else 0       # (the counter was reset before the timed steps)
    if distributed:
        tt = torch.tensor([dt], dtype=torch.float64, device=xdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        per = [torch.zeros(2, dtype=torch.int64, device=xdev) for _ in range(world)]
        dist.all_gather(per, torch.tensor([n_cand, skipped_per_step], dtype=torch.int64, device=xdev))
        cand_per_rank = [int(t[0].item()) for t in per]
        skipped_per_rank = [int(t[1].item()) for t in per]  # every rank's own count: the ranks hold different regions
        total_cand = sum(cand_per_rank)
        rccl_ranks = dist.get_world_size()                  # what the process group itself reports
    else:
        total_cand = n_cand
        cand_per_rank = [n_cand]
        skipped_per_rank = [skipped_per_step]
        rccl_ranks = 1
    # the metric's own multi-GPU config beside the headline family, at every N: all 200,000 exons of configs[3] cut `world` ways (collective calls: every rank)
    exome_strong = None
    if args.scale_base_regions > 0 and args.config == "practice62" and method == "svr" and not args.no_extras and not args.regions:
        exome_strong = exome_line(args, local_rank, stream, model_path, n_regions=args.scale_base_regions, rank=rank, world=world, distributed=distributed, xdev=xdev)

    if rank == 0:
        emitted, surv = acc.download_survivors()
        survivors_gathered = 0
        if distributed:
            survivors_gathered = sum(gather.sizes) // 24
            head = gather.recv[0][:gather.sizes[0]].cpu().numpy().view(capi.SURVIVOR_DTYPE)
            assert np.array_equal(head["cand_index"], surv["cand_index"]), "rank 0's own slice of the gather differs from its survivors"
        # with --dynamic-skip only the candidates that were scored count: the skipped candidates of EVERY rank are left out
        value = (total_cand - sum(skipped_per_rank)) * args.steps / dt
        k_ms = float(np.mean(kernel_ms))
        n_sv = acc.model_info()[0] if method == "svr" else 0
        alg_bytes = ALG_BYTES_PER_CAND * n_cand
        hbm = {"achieved": alg_bytes / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "algorithmic_bytes_per_launch": alg_bytes}
        hbm["frac"] = hbm["achieved"] / HBM_PEAK_GBS
        traffic, traffic_src = None, None
        tpath = newest_profile("r*_hbm_traffic.json")
        if traffic_run:
            traffic = traffic_run["bytes_per_launch"]
        elif tpath and args.config == "practice62" and method == "svr" and not distributed:
            try:
                traffic = json.load(open(tpath)).get("k_svr_dense_bytes_per_launch")
                traffic_src = os.path.relpath(tpath, ROOT)
            except Exception:
                traffic = None
        kern = "k_svr_dense" if method == "svr" else "k_logistic_dense"
        if method == "svr":
            ent = table_entries_min(P, grids)
            # FP64 operations the window-separable algorithm needs per support vector: 3 per candidate (multiply + FMA on table factors)
            # + ~45 per distinct table entry (window-sum arithmetic and one full-precision exp2)
            flops = float(n_sv) * (3.0 * n_cand + 45.0 * ent)
            if args.dynamic_skip and n_cand:
                flops *= float(n_cand - skipped_per_step) / float(n_cand)      # tiles that were left out did no work: neither candidates nor table entries
            ach = flops / (k_ms * 1e-3) / 1e12
            roof = {"bound": "fp64_valu", "achieved": ach, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_PEAK_TFLOPS,
                    "traffic": traffic, "kernel": kern, "kernel_ms": k_ms,
                    "algorithmic_flops_per_launch": flops, "table_entries_per_sv": ent,
                    "note": "the dense SVR kernel is bound by FP64 VALU issue + LDS, not by HBM (SURVEY.md section 8d); achieved = algorithmic FP64 flops of the "
                            "window-separable form / HIP-event kernel time; the HBM view of the same launch is in `hbm`",
                    "hbm": hbm}
        else:
            roof = {"bound": "hbm", "achieved": hbm["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm["frac"], "traffic": None,
                    "kernel": kern, "kernel_ms": k_ms, "algorithmic_bytes_per_launch": alg_bytes}
        roof["hbm_frac"] = hbm["frac"]; roof["hbm_achieved_gbs"] = hbm["achieved"]; roof["algorithmic_bytes_per_launch"] = alg_bytes      # (flat: see above)
        if traffic_run:
            roof["traffic_measured_in_this_run"] = True
            roof["traffic_detail"] = traffic_run
            roof["traffic_fetch_bytes"] = traffic_run.get("fetch_bytes_corrected"); roof["traffic_write_bytes"] = traffic_run.get("write_bytes")
            roof["traffic_note"] = ("WRITE_SIZE = the scores (1.05x: 456-byte rows, 64-byte granules); FETCH_SIZE x 2 = the records + the inputs (the model once per "
                                    "XCD's L2 = 14 MB, ~80 KB per tile of copy-table slices / bases / model rows fetched again); round 5's 2.4x was the "
                                    "lane-by-lane store path of the headline tile shape, not a profiler constant (tools/exp/traffic_abl.sh, ctx_probe.sh)")
        elif traffic_src:
            roof["traffic_from_profile"] = traffic_src
            roof["traffic_measured_in_this_run"] = False
        out = {
            "metric": f"candidate MIPs scored/sec ({'SVR' if method == 'svr' else 'logistic'})", "value": value, "unit": "candidates/s",
            "n_gpus": world, "rccl_ranks": rccl_ranks, "backend": args.backend if distributed else None, "shared_gpus": bool(args.share_gpus and distributed),
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "scale_family": (f"{args.config}, {args.scaling} scaling: `value` of every --gpus N line of this family is the same workload "
                             f"{'grown N-fold (the same work per GPU)' if args.scaling == 'weak' else 'cut N ways'}; the exome (configs[3]) is `exome_strong`"),
            "config": {"workload": desc, "n_sv": n_sv, "regions_rank0": len(regions), "dense_candidates_rank0": n_cand,
                       "dense_candidates_all_ranks": total_cand, "dense_candidates_per_rank": cand_per_rank, "result_windows_rank0": acc.window_count(),
                       "emitted_candidates_rank0": int(emitted.sum()), "survivors_rank0": int((surv["cand_index"] >= 0).sum()),
                       # reference-equivalent rate: candidates the reference would have constructed, per second (rank 0's share)
                       "emitted_candidates_per_s_rank0": float(emitted.sum()) * args.steps / dt,
                       "survivors_gathered_per_step": survivors_gathered,
                       "dynamic_skip": bool(args.dynamic_skip), "skipped_candidates_per_step_rank0": int(skipped_per_step),
                       "skipped_candidates_per_step_per_rank": skipped_per_rank,
                       "dense_candidates_covered_per_s": total_cand * args.steps / dt},
            "roofline": roof,
            "parity_checked": bool(gate), "parity_gate": gate,
            "kernels_ms": {kern: k_ms, "k_records": float(np.mean(records_ms)), "k_replay_condense(+memsets)": float(np.mean(replay_ms))},
        }
        # the driver's record keeps the SCALAR members of the standard objects (nested objects and lists are dropped there): the per-kernel times, the
        # full-size exome figure and the per-K rates ride along as flat keys of `roofline` / `config`, beside the nested objects other readers use
        out["roofline"]["kernels_ms"] = out["kernels_ms"]
        out["roofline"]["k_records_ms"] = out["kernels_ms"]["k_records"]
        out["roofline"]["k_replay_condense_ms"] = out["kernels_ms"]["k_replay_condense(+memsets)"]
        if exome_strong:
            exome_strong["compare_with"] = "exome_strong.value of the --gpus 1 line (the same BED on one GPU) - never with `value`, which is another workload"
            out["exome_strong"] = exome_strong
            # the same numbers inside `config` / `roofline`, short strings only: BASELINE configs[3] at full size, timed in this very run
            out["config"]["exome_full"] = {"workload": f"configs[3]: {exome_strong['regions']} exons, capture 150-170, SVR n_sv={n_sv}, {world} way(s)",
                                           "value": exome_strong["value"], "unit": "candidates/s", "seconds": exome_strong["seconds"],
                                           "dense_candidates": exome_strong["dense_candidates"], "regions": exome_strong["regions"], "n_gpus": world,
                                           "result_windows_rank0": exome_strong.get("result_windows_rank0")}
            for k in ("value", "seconds", "dense_candidates", "regions", "result_windows_rank0"):
                out["config"]["exome_full_" + k] = exome_strong.get(k)
            out["config"]["exome_full_workload"] = out["config"]["exome_full"]["workload"]
            if "roofline" in exome_strong:
                out["roofline"]["exome_full_frac"] = exome_strong["roofline"]["frac"]
                out["roofline"]["exome_full_achieved_tflops"] = exome_strong["roofline"]["achieved"]
                out["roofline"]["exome_full_table_entries_per_sv"] = exome_strong["roofline"]["table_entries_per_sv"]
                r = exome_strong["roofline"]
                out["roofline"]["exome_full"] = {"bound": r["bound"], "achieved": r["achieved"], "peak": r["peak"], "unit": r["unit"], "frac": r["frac"],
                                                 "table_entries_per_sv": r["table_entries_per_sv"], "hbm_frac": r["hbm"]["frac"],
                                                 "clock": "wall around the whole pass (records + k_svr_dense + replay/condense)"}
        if method == "svr":
            out["fp64"] = {"pairs_per_launch": n_cand * n_sv, "pairs_per_s": n_cand * n_sv / (k_ms * 1e-3),
                           "naive_equiv_tflops": n_cand * n_sv * 600.0 / (k_ms * 1e-3) / 1e12, "peak_tflops": FP64_PEAK_TFLOPS}
        ppath = newest_profile("r*_pmc.json")
        if ppath and args.config == "practice62" and method == "svr" and not distributed:
            try:
                c = json.load(open(ppath))["k_svr_dense"]
                cyc = c["GRBM_GUI_ACTIVE"] / 8.0                      # summed over the 8 XCDs
                out["pmc_from_profile"] = {"file": os.path.relpath(ppath, ROOT), "measured_in_this_run": False,
                                           "valu_instr_per_wave_pair": c["SQ_INSTS_VALU"] * 64.0 / (float(n_cand) * float(n_sv)),
                                           "valu_issue_frac": c["SQ_INSTS_VALU"] * 4.0 / (1024 * cyc), "lds_busy_frac": c["SQ_LDS_IDX_ACTIVE"] / (256 * cyc)}
            except Exception:
                pass
        # ---- extras (N = 1, default workload): throughput vs nSV, and the logistic scorer on the same batch -------------------------
        if not args.no_extras and not distributed and args.config == "practice62" and method == "svr":
            extra = []
            if exome_strong:
                # the N = 1 point of the exome's strong-scaling curve under its round-4 name as well
                out["scale_base"] = {"value": exome_strong["value"], "unit": exome_strong["unit"], "workload": exome_strong["what"], "seconds": exome_strong["seconds"],
                                     "dense_candidates": exome_strong["dense_candidates"], "regions": exome_strong["regions"], "n_gpus": 1, "scaling": "strong",
                                     "equivalent_command": f"python bench.py --gpus 1 --config exome --regions {args.scale_base_regions} --scaling strong",
                                     "roofline": exome_strong["roofline"]}
                by_k = exome_rates_by_sizes(args, local_rank, stream, model_path)
                extra.append({"what": "k_svr_dense on the exome regions that keep K capture sizes (mipgen.cpp:429), K by K: the first 4,096 exons",
                              "by_capture_sizes": by_k})
                out["config"]["exome_by_capture_sizes"] = {k: {"candidates_per_s": v["candidates_per_s"], "candidates_share": v["candidates_share"],
                                                               "regions": v["regions"]} for k, v in by_k.items()}
                for k, v in by_k.items():
                    out["config"][f"exome_K{k}_candidates_per_s"] = v["candidates_per_s"]
                    out["config"][f"exome_K{k}_candidates_share"] = v["candidates_share"]
            if args.sustain_seconds > 0:
                # the headline again, long enough for an outside observer (the driver's GPU-busy sampler) to see: same step, >= 2 s
                reps = max(args.steps, int(args.sustain_seconds / max(dt / args.steps, 1e-6)) + 1)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(reps):
                    step()
                torch.cuda.synchronize()
                d1 = time.perf_counter() - t1
                extra.append({"what": f"sustained: the headline step {reps} times back to back", "value": n_cand * reps / d1, "unit": "candidates/s",
                              "ms_per_step": d1 / reps * 1e3, "seconds": d1})
            # the same batch handed over as HOST buffers (the C ABI's mipgen_accel_score_regions): H2D of the region batch, dense scoring of every
            # candidate, D2H of 16 B per candidate - the PCIe-inclusive rate of the boundary (never the headline: inputs are resident there)
            acc.score_regions(regions, m)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                acc.score_regions(regions, m)
            torch.cuda.synchronize()
            d1 = time.perf_counter() - t1
            extra.append({"what": "PCIe-inclusive: mipgen_accel_score_regions from host buffers (upload of the region batch, dense scoring, download of "
                                  "score + record of every candidate)", "value": n_cand * reps / d1, "unit": "candidates/s", "ms_per_step": d1 / reps * 1e3,
                          "bytes_down_per_step": 16 * n_cand})
            acc.upload(regions)                               # back to the resident-batch state of the other lines
            if args.exome_regions > 0:
                extra.append(exome_line(args, local_rank, stream, model_path))
            for nsv in (256, 4096):
                acc.load_model_file(workloads.svr_model_path(cache, model_genome, nsv, rho=MODEL_RHO))
                acc.score_condense_all(capi.SCORE_SVR)
                torch.cuda.synchronize()
                reps = 3
                t1 = time.perf_counter()
                ks = []
                for _ in range(reps):
                    acc.score_condense_all(capi.SCORE_SVR)
                    ks.append(acc.last_kernel_ms(0))
                torch.cuda.synchronize()
                d1 = time.perf_counter() - t1
                extra.append({"what": f"same batch, n_sv={nsv}", "value": n_cand * reps / d1, "unit": "candidates/s", "ms_per_step": d1 / reps * 1e3,
                              "k_svr_dense_ms": float(np.mean(ks)), "pairs_per_s": n_cand * nsv / (float(np.mean(ks)) * 1e-3)})
            acc.score_condense_all(capi.SCORE_LOGISTIC)
            torch.cuda.synchronize()
            reps = 10
            t1 = time.perf_counter()
            ks = []
            for _ in range(reps):
                acc.score_condense_all(capi.SCORE_LOGISTIC)
                ks.append(acc.last_kernel_ms(2))
            torch.cuda.synchronize()
            d1 = time.perf_counter() - t1
            lk = float(np.mean(ks))
            extra.append({"what": "same batch, logistic scoring (k_logistic_dense + replay/condense)", "value": n_cand * reps / d1, "unit": "candidates/s",
                          "ms_per_step": d1 / reps * 1e3, "k_records_logistic_ms": lk,
                          "roofline": {"bound": "hbm", "achieved": alg_bytes / (lk * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": alg_bytes / (lk * 1e-3) / 1e9 / HBM_PEAK_GBS}})
            # mixed designs (BASELINE configs[2]): the condensed survivors of the logistic scan re-scored by the SVR as ONE list - features per
            # candidate (k_features_batch), then all candidate x SV distances on the FP64 matrix cores (k_svr_gemm)
            acc.load_model_file(model_path)
            _, lsurv = acc.download_survivors()
            A = P.n_arm_pairs
            lc = []
            pos = 0
            for ri, gr in enumerate(acc.grids):
                sv = lsurv[2 * pos:2 * (pos + gr.n_pos)]
                idx = sv["cand_index"][sv["cand_index"] >= 0] - gr.offset
                a_ = idx % A; row = idx // A; st = row & 1; rest = row >> 1; ki = rest % gr.n_sizes; pi = rest // gr.n_sizes
                lc += [(ri, gr.first_pos + int(pi[k]), P.max_capture_size - (gr.first_size_index + int(ki[k])) * P.capture_increment,
                        P.arm_ext[int(a_[k])], P.arm_lig[int(a_[k])], int(st[k])) for k in range(idx.shape[0])]
                pos += gr.n_pos
            if len(lc) >= 256:
                acc.score_candidates(lc, capi.SCORE_SVR)
                acc.score_candidates(lc, capi.SCORE_SVR)
                gms, fms = acc.last_kernel_ms(5), acc.last_kernel_ms(6)
                nsv_l = acc.model_info()[0]
                fl = 2.0 * len(lc) * ((nsv_l + 63) // 64 * 64) * 192
                extra.append({"what": f"mixed-mode re-scoring of the {len(lc)} condensed survivors of the logistic scan as one list (n_sv={nsv_l})",
                              "value": len(lc) / ((fms + gms) * 1e-3), "unit": "re-scored candidates/s (the two kernels; HIP events)",
                              "k_features_batch_ms": fms, "k_svr_gemm_ms": gms,
                              "roofline": {"bound": "mfma", "achieved": fl / (gms * 1e-3) / 1e12, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                           "frac": fl / (gms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, "kernel": "k_svr_gemm",
                                           "note": "v_mfma_f64_16x16x4f64 flops of the candidate x SV products / HIP-event kernel time; the FP64 matrix "
                                                   "peak of MI355X equals its vector peak (78.6 TFLOP/s); the kernel also takes one exp2 per pair"}})
            # arm-oligo copy numbers by exact k-mer counting - one streaming pass over the genome (1 B / base)
            from mipgen_amd import synth
            gsz = 64 << 20
            big = synth.random_genome(gsz, 77)
            lens = sorted({e for e, _ in capi.arm_pairs_of(P)} | {l for _, l in capi.arm_pairs_of(P)})
            acc.count_oligo_copies([big], [rd.seq for rd in regions], lens)
            acc.count_oligo_copies([big], [rd.seq for rd in regions], lens)
            kms = acc.last_kernel_ms(4)
            extra.append({"what": f"oligo copy numbers without bwa: exact k-mer counting of the {len(regions)} region strings x {len(lens)} oligo lengths against a "
                                  f"{gsz >> 20} MiB genome (k_kmer_count: genome streamed once, one Bloom-filter bit test per base, table probes only for the survivors)", "k_kmer_count_ms": kms,
                          "roofline": {"bound": "hbm", "achieved": gsz / (kms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gsz / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                       "algorithmic_bytes_per_launch": gsz}})
            out["extra"] = extra
            acc.load_model_file(model_path)
        if not args.no_cpu_baseline and not distributed:
            mp = model_path or workloads.svr_model_path(cache, model_genome, args.nsv, rho=MODEL_RHO)
            try:
                out["cpu_baseline"] = cpu_baseline(args, mp, args.nsv)
            except Exception as e:                          # the baseline leg must never take the measured line down with it
                out["cpu_baseline"] = {"value": None, "unit": "candidates/s", "cores": 0, "kind": "reference", "sample": "failed", "error": repr(e)[:400]}
        print(json.dumps(out))
    acc.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

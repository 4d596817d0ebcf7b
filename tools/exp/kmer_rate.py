# tools/exp/kmer_rate.py: k_kmer_count's genome pass at several genome sizes (is the 64 MiB figure of bench.py launch-bound?)
import os, sys, numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R)
from mipgen_amd import capi, synth, workloads
genome, ivs = workloads.practice62()
P = capi.make_params(140, 180, score_method=capi.SCORE_LOGISTIC)
acc = capi.Accel(P)
regions = workloads.build_regions(acc, genome, ivs, P)
lens = sorted({e for e, _ in capi.arm_pairs_of(P)} | {l for _, l in capi.arm_pairs_of(P)})
for mb in (64, 256, 1024):
    big = synth.random_genome(mb << 20, 77)
    acc.count_oligo_copies([big], [rd.seq for rd in regions], lens)
    acc.count_oligo_copies([big], [rd.seq for rd in regions], lens)
    ms = acc.last_kernel_ms(4)
    print(f"{mb} MiB genome: k_kmer_count {ms:.3f} ms = {(mb << 20) / (ms * 1e-3) / 1e9:.0f} GB/s = {(mb << 20) / (ms * 1e-3) / 8e12 * 100:.1f} % of 8 TB/s", flush=True)

# tools/exp/logistic_time.py [LIB]: kernel times of the dense logistic path on 24 x 5 kb regions (capture 120-250), for A/B runs of scratch builds
import os, sys, hashlib, numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
from mipgen_amd import capi, workloads
lib_path = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] != "-" else None
cfg = sys.argv[2] if len(sys.argv) > 2 else "regions5k"
if lib_path:
    capi.LIB_PATH = os.path.join(R, lib_path); capi._lib = capi.load_library(capi.LIB_PATH)
if cfg == "regions5k":
    P = capi.make_params(120, 250, score_method=capi.SCORE_LOGISTIC)
    acc = capi.Accel(P)
    ivs = workloads.regions5k_intervals(24)
    regions = workloads.build_regions5k(acc, workloads.regions5k_genome(), ivs, P, with_lrc=False)
else:                                              # "exome": 8,192 exons, capture 150-170 (k_replay_condense_carry<5>)
    chrom_len, all_iv = workloads.exome_layout()
    P = capi.make_params(150, 170, score_method=capi.SCORE_LOGISTIC)
    acc = capi.Accel(P)
    regions = workloads.build_exome(acc, chrom_len, all_iv[:8192], P)
acc.upload(regions)
acc.set_timing(True)
ts = []
for _ in range(14):
    acc.score_window(0, capi.SCORE_LOGISTIC); t2 = acc.last_kernel_ms(2)
    acc.replay_condense(); ts.append([t2, acc.last_kernel_ms(3)])
ts = np.array(ts[3:])
sc, rec = acc.download()
em, surv, _ = acc.download_replay(want_mask=False)
chk = hashlib.md5(np.ascontiguousarray(sc).tobytes()).hexdigest()[:12] + hashlib.md5(np.ascontiguousarray(rec).tobytes()).hexdigest()[:8]
chk2 = hashlib.md5(np.ascontiguousarray(em).tobytes()).hexdigest()[:8] + hashlib.md5(np.ascontiguousarray(surv).tobytes()).hexdigest()[:8]
print(f"{cfg} logistic {lib_path or 'product'}: [k_logistic_dense, replay + condense] ms min {ts.min(axis=0).round(3).tolist()} median {np.median(ts, axis=0).round(3).tolist()} checksum {chk} replay {chk2}")

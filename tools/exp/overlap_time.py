# tools/exp/overlap_time.py: do k_logistic_dense (latency / issue bound, 31 % of HBM) and the replay + condense kernel (HBM bound) of ANOTHER window run
# faster side by side than one after the other?  Two handles with their own streams on the same 24 x 5 kb batch stand in for two result windows.
import os, sys, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
import torch
from mipgen_amd import capi, workloads
P = capi.make_params(120, 250, score_method=capi.SCORE_LOGISTIC)
ivs = workloads.regions5k_intervals(24)
genome = workloads.regions5k_genome()
accs = []
for k in range(2):
    a = capi.Accel(P)
    a.upload(workloads.build_regions5k(a, genome, ivs, P, with_lrc=False))
    accs.append(a)
def sync(): torch.cuda.synchronize()
N = 40
for a in accs: a.score_window(0, capi.SCORE_LOGISTIC); a.replay_condense()
sync()
t0 = time.perf_counter()
for i in range(N):
    a = accs[i & 1]
    a.score_window(0, capi.SCORE_LOGISTIC); a.replay_condense()
    sync()
t_seq = (time.perf_counter() - t0) / N
# pipelined: the dense kernel of window i + 1 is enqueued (other handle, other stream) before the replay of window i
accs[0].score_window(0, capi.SCORE_LOGISTIC); sync()
t0 = time.perf_counter()
for i in range(N):
    cur, nxt = accs[i & 1], accs[(i + 1) & 1]
    nxt.score_window(0, capi.SCORE_LOGISTIC)
    cur.replay_condense()
    sync()
t_pipe = (time.perf_counter() - t0) / N
print(f"per window: one after the other {t_seq * 1e3:.3f} ms, dense(i+1) beside replay(i) {t_pipe * 1e3:.3f} ms")

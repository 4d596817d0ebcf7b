"""Worker for tests/test_host_select_cpu.py::test_two_ranks_gather_survivors_and_rank0_picks (gloo, CPU): the multi-rank design flow of
mipgen_amd/dist.py with REAL survivor records - shard by dense-grid size, score the shard (oracle), one gather of the survivors to rank 0,
sequential pick on rank 0 through libmipgen_host.so."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch.distributed as dist  # noqa: E402

from mipgen_amd import capi, dist as mdist  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from tests import helpers as H  # noqa: E402
from tests import host_select_common as HS  # noqa: E402


def main() -> None:
    name, base, out_path = sys.argv[1], sys.argv[2], sys.argv[3]
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    meta = H.load_design(name)
    work = os.path.join(base, f"rank{rank}")
    os.makedirs(work)
    os.environ["FAKEBWA_MODE"] = meta["bwa"]
    d = HS.open_design(H.prepare_cli_workdir(meta, work) + ["-silent_mode", "on"], work)      # every rank runs the (cheap) input stage
    P = d.params()
    views = HS.design_views(d)
    grids = [po.grid(P, v) for v in views]
    shards = mdist.shard_regions([g.count for g in grids], world)
    lo, hi = shards[rank]
    scan = capi.SCORE_SVR if d.score_method == capi.SCORE_SVR else capi.SCORE_LOGISTIC
    model = po.Model(d.model_path) if d.score_method != capi.SCORE_LOGISTIC else None
    local = [HS.oracle_region_results(P, views[i], scan, model) for i in range(lo, hi)]
    surv = np.concatenate([r["survivors"] for r in local]) if local else np.zeros(0, dtype=capi.SURVIVOR_DTYPE)
    emitted = np.array([r["emitted"] for r in local], dtype=np.int64)
    all_surv = mdist.gather_to_rank0(surv)                      # the one exchange step of the path
    all_emitted = mdist.gather_to_rank0(emitted)
    if rank == 0:
        pos = 0
        rescore = HS.make_rescorer(P, views, model) if d.score_method == capi.SCORE_MIXED else None
        with HS.in_dir(work):
            for i, g in enumerate(grids):
                d.select_region(i, g, all_surv[2 * pos:2 * (pos + g.n_pos)], int(all_emitted[i]), rescore=rescore)
                pos += g.n_pos
        c = d.counters()
        json.dump({"shards": shards, "n_regions": len(grids), "picked": c["picked"], "all_mips": c["all_mips"]}, open(out_path, "w"))
    d.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

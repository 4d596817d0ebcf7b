"""Worker for tests/test_gpu_sharded.py::test_rccl_collectives_of_the_multi_gpu_paths_with_one_rank: the torch.distributed calls bench.py --gpus N and
mipgen_amd/mp_design.py make - backend "nccl" (= RCCL), CUDA tensors of the same dtypes and shapes - on a process group of ONE rank, so that the first
execution of these calls on RCCL is not the driver's multi-GPU run."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from mipgen_amd import capi, dist as mdist  # noqa: E402


def main() -> None:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29547")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
    assert dist.get_world_size() == 1 and dist.get_backend() == "nccl"
    dev = "cuda:0"
    # mipgen_amd/dist.py: the padded byte gather of structured survivors, and the exclusive scan of record counts
    surv = np.zeros(1000, dtype=capi.SURVIVOR_DTYPE)
    surv["cand_index"] = np.arange(1000); surv["score"] = np.linspace(0, 1, 1000); surv["record"] = np.arange(1000, dtype=np.uint64) * 7
    got = mdist.gather_to_rank0(surv, dev)
    assert got.dtype == surv.dtype and np.array_equal(got, surv)
    assert np.array_equal(mdist.gather_to_rank0(np.arange(17, dtype=np.int32), dev), np.arange(17, dtype=np.int32))
    assert mdist.gather_to_rank0(np.zeros(0, dtype=np.float64), dev).shape[0] == 0                 # an empty shard
    assert mdist.exclusive_offsets(1234, dev) == (0, 1234)
    # bench.py: size exchange, the per-step gather of the library's survivor bytes, max-over-ranks clock, per-rank counts, barriers
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev)]
    dist.all_gather(sizes, torch.tensor([24000], dtype=torch.int64, device=dev))
    assert int(sizes[0].item()) == 24000
    pad = torch.arange(24000, dtype=torch.int64, device=dev).to(torch.uint8)
    recv = [torch.zeros(24000, dtype=torch.uint8, device=dev)]
    dist.gather(pad, recv, dst=0)
    assert torch.equal(recv[0], pad)
    t = torch.tensor([3.25], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t.item()) == 3.25
    ok = torch.tensor([1], dtype=torch.int32, device=dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)                                                       # mp_design: "every shard scored"
    assert int(ok.item()) == 1
    dist.barrier()
    torch.cuda.synchronize()
    dist.destroy_process_group()
    print("rccl one-rank collectives ok")


if __name__ == "__main__":
    main()

"""Can RCCL run several ranks on ONE device (so that the multi-rank paths could meet RCCL on a 1-GPU box)?
usage: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/try_rccl_one_gpu.py"""
import os
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    t = torch.full((1 << 20,), float(dist.get_rank() + 1), device="cuda")
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print("rank", dist.get_rank(), "all_reduce ok:", float(t[0]))
    a = torch.arange(8, device="cuda", dtype=torch.float32) + 10 * dist.get_rank()
    b = torch.empty_like(a)
    dist.all_to_all_single(b, a)
    torch.cuda.synchronize()
    print("rank", dist.get_rank(), "all_to_all_single ok:", b.tolist())
    dist.destroy_process_group()
except Exception as e:
    print("rank", os.environ.get("RANK"), "FAILED:", repr(e)[:300])

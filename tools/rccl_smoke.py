"""RCCL sanity on a one-GPU box: world size 1, the bench's collective on a tensor the engine wrote"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch, torch.distributed as dist
import apsu_amd                                             # loads the HIP runtime the way bench.py does
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
x = torch.arange(2 * 8192, dtype=torch.int64, device="cuda").reshape(1, 2, 8192)
g = torch.zeros_like(x)
dist.all_gather_into_tensor(g, x)
dist.barrier(); torch.cuda.synchronize()
assert (g == x).all()
t = torch.tensor([1.5], dtype=torch.float64, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX)
print("rccl ok", float(t.item()))
dist.destroy_process_group()

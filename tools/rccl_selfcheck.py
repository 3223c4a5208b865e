"""RCCL sanity on a one-GPU box: a one-rank "nccl" process group running the collective pattern of bench.py's N>1 path
(all_gather_into_tensor ordered after an external stream through events, barrier, device synchronise).  Multi-rank RCCL
needs one GPU per rank and cannot be rehearsed here; this checks that the library loads and the call sequence is legal."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch, torch.distributed as dist
import apsu_amd
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
js = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "params", "1M-1024-com.json")).read()
ctx = apsu_amd.HeContext(js, device=0)
ext = torch.cuda.ExternalStream(ctx.stream, device=dev)
out = [torch.arange(4 * 2 * ctx.n, dtype=torch.int64, device=dev).reshape(4, 2, ctx.n) + k for k in range(2)]
gathered = torch.zeros((4, 2, ctx.n), dtype=torch.int64, device=dev)
free = [None, None]
for step in range(6):
    slot = step & 1
    if free[slot] is not None:
        ext.wait_event(free[slot])
    with torch.cuda.stream(ext):
        out[slot].add_(1)                                   # stands in for the engine's queued work on its own stream
    cur = torch.cuda.current_stream()
    cur.wait_stream(ext)
    dist.all_gather_into_tensor(gathered, out[slot])
    free[slot] = torch.cuda.Event(); free[slot].record(cur)
dist.barrier()
torch.cuda.synchronize()
want = torch.arange(4 * 2 * ctx.n, dtype=torch.int64, device=dev).reshape(4, 2, ctx.n) + 1 + 3
assert bool((gathered == want).all()), "gathered rows differ"
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t.item()) == 1.5
dist.destroy_process_group()
print("rccl one-rank selfcheck ok")

"""CPU tier, world_size 2 over gloo: the BinBundle sharding + result gather used by bench.py
(apsu_amd/sharding.py).  Each rank evaluates its shard with the CPU oracle (stand-in for the engine,
which needs a GPU), results are all-gathered like the RCCL path does, and rank 0 checks every
BinBundle's result against a single-process evaluation."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import common
from apsu_amd.sharding import gather_slots, partition


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, degrees, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        S = common.make_scenario(common.toy_json(), degrees)
        units = [(b["bundle_idx"], b["cache_idx"], b["degree"]) for b in S.bundles]
        assign = partition(units, S.p["bundle_idx_count"], world)
        max_local, rows = gather_slots(assign)
        mine = assign[rank]
        my_idx = sorted({u[0] for u in mine})
        pw = {b: S.C.compute_powers(S.src[b], S.nodes, S.rk, S.ps_low) for b in my_idx}   # only this rank's indices
        out = torch.zeros((max_local, 2, S.C.n), dtype=torch.int64)
        by_key = {(b["bundle_idx"], b["cache_idx"]): b for b in S.bundles}
        for i, u in enumerate(mine):
            res = common.oracle_eval(S, pw, by_key[(u[0], u[1])])
            out[i] = torch.from_numpy(res.reshape(2, -1).view(np.int64))
        gathered = torch.zeros((world * max_local, 2, S.C.n), dtype=torch.int64)
        dist.all_gather_into_tensor(gathered, out)
        if rank == 0:
            full = common.oracle_powers(S)
            ok = True
            for b in S.bundles:
                exp = common.oracle_eval(S, full, b).reshape(2, -1)
                got = gathered[rows[(b["bundle_idx"], b["cache_idx"])]].numpy().view(np.uint64)
                ok = ok and bool((got == exp).all())
            q.put(ok)
    finally:
        dist.destroy_process_group()


def test_two_rank_shard_and_gather():
    degrees = {0: [10, 3, 11, 7], 1: [11, 4, 9]}
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, degrees, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_partition_covers_every_binbundle_once(world):
    # the 16M-4096 synthetic DB of bench.py: 4 indices x (6 full + 1 short)
    units = [(b, ci, 1303 if ci < 6 else 170) for b in range(4) for ci in range(7)]
    assign = partition(units, 4, world)
    flat = [u for r in range(world) for u in assign[r]]
    assert sorted(flat) == sorted(units)
    # a rank only ever needs the powers of few bundle indices
    for r in range(world):
        assert len({u[0] for u in assign[r]}) <= max(1, 4 // world)
    loads = [sum(u[2] for u in assign[r]) for r in range(world)]
    assert max(loads) <= 1.35 * (sum(loads) / world) + 1303
    max_local, rows = gather_slots(assign)
    assert len(set(rows.values())) == len(units) and max(rows.values()) < world * max_local

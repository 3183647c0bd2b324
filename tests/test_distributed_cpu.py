"""The N>1 host path on CPU: world_size-2 gloo processes broadcast the weight blob, shard the
instances and merge their result dictionaries (no GPU, no data-path collective)."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

from diffusionsat_amd import distributed as D  # noqa: E402
from diffusionsat_amd.instances import planted_ksat  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        blob = torch.arange(1000, dtype=torch.float32) if rank == 0 else torch.zeros(1000)
        D.broadcast_weight_blob(blob, src=0)
        insts = [planted_ksat(10 + i, 20 + 3 * i, 3, seed=i)[:2] for i in range(7)]
        mine = D.shard_instances(insts, rank, world)
        local = {(1 << 70) + rank: 2, 5: 1 + rank}
        merged = D.merge_counts(local)
        q.put((rank, float(blob.sum()), [n for n, _ in mine], {str(k): v for k, v in merged.items()},
               D.split_sample_budget(7, rank, world)))
    finally:
        dist.destroy_process_group()


def test_world_size_two_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(abs(r[1] - 499500.0) < 1e-3 for r in res)            # broadcast reached both ranks
    all_n = sorted(res[0][2] + res[1][2])
    assert all_n == sorted(10 + i for i in range(7))                  # every formula on exactly one rank
    assert abs(len(res[0][2]) - len(res[1][2])) <= 1
    want = {str((1 << 70)): 2, str((1 << 70) + 1): 2, "5": 3}
    assert res[0][3] == want and res[1][3] == want                    # big-int keys survive the merge
    assert res[0][4] + res[1][4] == 7


def test_single_process_helpers():
    assert D.merge_counts({3: 1}) == {3: 1}
    assert [D.split_sample_budget(10, r, 4) for r in range(4)] == [3, 3, 2, 2]
    insts = [(5, [[1]] * 9), (50, [[1]] * 200), (6, [[1]] * 3)]
    a, b = D.shard_instances(insts, 0, 2), D.shard_instances(insts, 1, 2)
    assert len(a) + len(b) == 3 and (insts[1] in a) != (insts[1] in b)

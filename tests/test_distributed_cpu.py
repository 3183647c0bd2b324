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


class _FakeDeviceSampler:
    """Mixin that replaces the two device hooks of DiffusionSampler (``_launch`` / ``_collect``) by a
    host stand-in: random bits from the batch's seed, per-copy SAT flags from the formula.  Everything
    else -- budget split, Philox sub-stream seeds, decoding, abort rule, merge -- is the product code."""

    def _launch(self, it, copies, seed):
        return copies, seed

    def _collect(self, handle):
        from diffusionsat_amd.dimacs import VariableAssignment
        copies, seed = handle
        bits = np.random.default_rng(seed).integers(0, 2, copies * self.n_vars).astype(np.uint8)
        va = VariableAssignment(n_vars=self.n_vars, clauses=self.clauses)
        solved = np.zeros(copies, bool)
        for i in range(copies):
            va.assign_all_from_bit_list(bits[i * self.n_vars:(i + 1) * self.n_vars])
            solved[i] = va.satisfiable()
        packed = np.packbits(bits.reshape(copies, self.n_vars), axis=1, bitorder="little")
        return packed.reshape(-1), solved


def _make_fake_sampler(path, rank, world):
    from diffusionsat_amd.diffusion_sampler import DiffusionSampler

    class S(_FakeDeviceSampler, DiffusionSampler):
        pass
    return S(None, path, device="cuda:0", seed=5, rank=rank, world_size=world, copies_per_batch=7)


def _sampler_worker(rank, world, port, path, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        smp = _make_fake_sampler(path, rank, world)
        res = smp.samples(23)
        q.put((rank, {str(k): v for k, v in res.items()}, smp.success_rate))
    finally:
        dist.destroy_process_group()


def test_diffusion_sampler_samples_world_size_two(tmp_path):
    """DiffusionSampler.samples with world_size 2 over gloo: every rank draws its share of the budget
    from its own seeds and both return the same merged dictionary = the union of what each rank's
    local loop produces (recomputed here in one process with the same stand-in device hooks)."""
    from diffusionsat_amd.dimacs import DimacsFile
    cnf = tmp_path / "f.cnf"
    DimacsFile(str(cnf), clauses=[[1, 2], [-2, 3], [3, 4, -5], [-1, -4]]).store()
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sampler_worker, args=(r, world, port, str(cnf), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] and res[0][2] == res[1][2]          # both ranks hold the merged result
    want, tot, sat = {}, 0, 0
    for r in range(world):
        local, t, s = _make_fake_sampler(str(cnf), r, world)._samples_local(D.split_sample_budget(23, r, world))
        assert sum(local.values()) == D.split_sample_budget(23, r, world)
        tot, sat = tot + t, sat + s
        for k, v in local.items():
            want[str(k)] = want.get(str(k), 0) + v
    assert res[0][1] == want and sum(want.values()) == 23
    assert abs(res[0][2] - sat / tot) < 1e-12
    # world_size > 1 without an initialised process group is an error, not a silent single-rank run
    with pytest.raises(RuntimeError):
        _make_fake_sampler(str(cnf), 0, 2).samples(3)


def test_epoch_batch_sizes_follow_the_reference_batcher():
    from diffusionsat_amd.diffusion_sampler import epoch_batch_sizes
    assert epoch_batch_sizes(10, 4) == [4, 4]                 # copies 5 and 10 overflow a batch and are lost (F11)
    assert epoch_batch_sizes(10, 4, reference_drop_bug=False) == [4, 4, 2]
    assert epoch_batch_sizes(3, 4) == [3] and epoch_batch_sizes(0, 4) == []
    assert epoch_batch_sizes(3, 1) == [1, 1]                  # a formula larger than the budget: alone in its batch


def test_single_process_helpers():
    assert D.merge_counts({3: 1}) == {3: 1}
    assert [D.split_sample_budget(10, r, 4) for r in range(4)] == [3, 3, 2, 2]
    insts = [(5, [[1]] * 9), (50, [[1]] * 200), (6, [[1]] * 3)]
    a, b = D.shard_instances(insts, 0, 2), D.shard_instances(insts, 1, 2)
    assert len(a) + len(b) == 3 and (insts[1] in a) != (insts[1] in b)

"""world_size-2 gloo test of the multi-GPU sharding path (no GPU: the CPU oracle stands in
for the per-rank scorer, which is allowed in tests only)."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from nanorepeat_amd import dist as D, synth
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        d = synth.make_1d(9, "TATTG", (6, 13), "ont_q20", kwin=(2, 17), anchor=150, flank=60, seed=4)
        out = D.round3_1d_sharded(d["regions"], d["reads"], d["kmin"], d["kmax"],
                                  scorer=lambda *a, **k: O.round3_1d(*a, threads=1, **k))
        q.put((rank, {k: v.tolist() for k, v in out.items()}))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_equals_single_process(oracle):
    from nanorepeat_amd import synth
    world, port = 2, 29500 + os.getpid() % 2000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    d = synth.make_1d(9, "TATTG", (6, 13), "ont_q20", kwin=(2, 17), anchor=150, flank=60, seed=4)
    want = oracle.round3_1d(d["regions"], d["reads"], d["kmin"], d["kmax"])
    for r in range(world):
        for k in ("best_score", "sum_k", "n_ties", "status"):
            assert res[r][k] == want[k].tolist(), (r, k)


def test_shard_reads_is_a_balanced_partition():
    from nanorepeat_amd import dist as D
    rng = np.random.default_rng(0)
    cost = rng.integers(1, 1000, size=501)
    for world in (1, 2, 3, 8):
        shards = D.shard_reads(cost, world)
        allidx = np.sort(np.concatenate(shards))
        assert np.array_equal(allidx, np.arange(len(cost)))
        loads = [int(cost[s].sum()) for s in shards]
        assert max(loads) - min(loads) <= cost.max()


def test_estimate_cells_formula():
    from nanorepeat_amd import dist as D
    regions = [("A" * 10, "CAG", "T" * 7)]
    cells = D.estimate_cells(regions, ["ACGTA", ""], [2, 0], [4, -1])
    assert cells.tolist() == [5 * sum(17 + 3 * k for k in (2, 3, 4)), 0]


def _gpu_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from nanorepeat_amd import dist as D, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        d = synth.make_1d(40, "TATTG", (9, 27), "ont", kwin=None, anchor=300, seed=14)
        out = D.round3_1d_sharded(d["regions"], d["reads"], d["kmin"], d["kmax"], device=0)   # both ranks share GPU 0
        q.put((rank, {k: v.tolist() for k, v in out.items()}))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_sharded_hip_path_two_ranks_one_gpu(capi, oracle):
    """The N > 1 path with the real HIP scorer: two spawned ranks (gloo rendezvous, both on GPU 0)
    each score their shard through the C ABI; every rank ends with the single-process answer."""
    from nanorepeat_amd import synth
    world, port = 2, 31500 + os.getpid() % 2000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gpu_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    d = synth.make_1d(40, "TATTG", (9, 27), "ont", kwin=None, anchor=300, seed=14)
    want = oracle.round3_1d(d["regions"], d["reads"], d["kmin"], d["kmax"])
    for r in range(world):
        for k in ("best_score", "sum_k", "n_ties", "status"):
            assert res[r][k] == want[k].tolist(), (r, k)

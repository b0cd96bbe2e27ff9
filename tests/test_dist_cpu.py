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


def _worker_2d(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from nanorepeat_amd import dist as D, synth
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        j, cr, k1, k2 = _joint_case(synth)
        out = D.joint_2d_sharded(j["region"], j["reads"], cr, k1, k2,
                                 scorer=lambda *a, **k: O.joint_2d(*a, threads=1, **{x: y for x, y in k.items() if x != "device"}))
        q.put((rank, {k: v.tolist() for k, v in out.items()}))
    finally:
        dist.destroy_process_group()


def _joint_case(synth):
    j = synth.make_joint(7, alleles=((6, 4), (11, 3)), read_len=420, read_sd=25, anchor=200, seed=19)
    cr, k1, k2 = [], [], []
    for r in range(7):
        for a in range(3, 14, 2 + r % 2):
            for b in range(1, 7, 2):
                cr.append(r); k1.append(a); k2.append(b)
    return j, cr, k1, k2


@pytest.mark.timeout(300)
def test_joint_sharded_equals_single_process(oracle):
    """2D: reads sharded over two ranks, every rank runs the full grid of its own reads."""
    from nanorepeat_amd import synth, dist as D
    world, port = 2, 33500 + os.getpid() % 2000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_2d, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    j, cr, k1, k2 = _joint_case(synth)
    want = oracle.joint_2d(j["region"], j["reads"], cr, k1, k2)
    for r in range(world):
        for k in ("read_strand", "best_wscore", "sum_k1", "sum_k2", "n_ties", "status"):
            assert res[r][k] == want[k].tolist(), (r, k)
    cells = D.estimate_cells_2d(j["region"], j["reads"], cr, k1, k2)
    assert cells.shape == (7,) and (cells > 0).all()
    # world size 1 (no process group): same answer, no collective
    one = D.joint_2d_sharded(j["region"], j["reads"], cr, k1, k2, scorer=lambda *a, **k: oracle.joint_2d(*a, **{x: y for x, y in k.items() if x != "device"}))
    for k in ("read_strand", "best_wscore", "sum_k1", "sum_k2", "n_ties", "status"):
        assert one[k].tolist() == want[k].tolist(), k


def _worker_regions(rank, world, port, q, fail_rank):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from nanorepeat_amd import dist as D, synth
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        d = synth.config4(5, 6, seed=11)

        def scorer(*a, **k):
            if rank == fail_rank:
                raise ValueError("boom")
            return O.round3_1d(*a, threads=1, **k)
        try:
            out = D.round3_1d_sharded(d["regions"], d["reads"], d["kmin"], d["kmax"], read_region=d["read_region"],
                                      scorer=scorer)
            q.put((rank, {k: v.tolist() for k, v in out.items()}))
        except RuntimeError as e:
            q.put((rank, "error: " + str(e)))
    finally:
        dist.destroy_process_group()


def _run_regions(fail_rank):
    world, port = 2, 37500 + os.getpid() % 2000 + (0 if fail_rank < 0 else 7)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_regions, args=(r, world, port, q, fail_rank)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


@pytest.mark.timeout(300)
def test_region_block_sharding_equals_single_process(oracle):
    """Many regions: region blocks dealt by executed cells over two ranks (gloo) == one process."""
    from nanorepeat_amd import synth
    res = _run_regions(-1)
    d = synth.config4(5, 6, seed=11)
    want = oracle.round3_1d(d["regions"], d["reads"], d["kmin"], d["kmax"], read_region=d["read_region"])
    for r in range(2):
        for k in ("best_score", "sum_k", "n_ties", "status"):
            assert res[r][k] == want[k].tolist(), (r, k)


@pytest.mark.timeout(300)
def test_scorer_failure_on_one_rank_raises_on_all_ranks():
    """A rank whose scorer raises still joins the all_gather; every rank then raises (no hang)."""
    res = _run_regions(1)
    assert all(isinstance(v, str) and "rank(s) [1]" in v for v in res.values()), res


def test_region_block_sharder_properties():
    from nanorepeat_amd import dist as D
    rng = np.random.default_rng(3)
    # 40 regions of very different sizes, reads interleaved in the input
    rr = rng.integers(0, 40, size=5000)
    cost = rng.integers(1000, 9000, size=5000) * (1 + rr % 5)
    for world in (1, 2, 4, 8):
        owner = D.shard_region_blocks(cost, rr, world)
        assert owner.shape == (5000,) and owner.min() >= 0 and owner.max() < world
        loads = np.bincount(owner, weights=cost, minlength=world)
        assert loads.max() <= 1.15 * loads.mean()
        # a region is cut into few blocks: its reads land on few ranks, in input-order runs
        for g in (0, 7, 39):
            o = owner[rr == g]
            assert (np.diff(o) != 0).sum() <= 2 * world
    # one region, many ranks (config 2 strong scaling): still balanced
    owner = D.shard_region_blocks(np.full(1000, 7), None, 8)
    assert np.bincount(owner, minlength=8).min() >= 100
    # executed-cell cost: padded rows x executed columns (csrc/nra_host.cpp: half-wave sweeps up to 768 bases with a
    # pipeline 31 lanes deep, full-wave ones up to 1536 with 63, row blocks of 64 x 12 .. 15 rows beyond -- the height that
    # pads least; the forward pipeline is skewed by the unit length)
    cells = D.executed_cells([("A" * 10, "CAG", "T" * 7)], [5, 65, 769, 1536, 1537, 2049, 3073], [4] * 7, fold=False)
    assert cells.tolist() == [32 * (17 + 12 + 31 * 4), 96 * (17 + 12 + 31 * 4), 64 * 13 * (17 + 12 + 63 * 4),
                              64 * 24 * (17 + 12 + 63 * 4), 2 * 832 * (17 + 12 + 63 * 4), 3 * 768 * (17 + 12 + 63 * 4),
                              4 * 832 * (17 + 12 + 63 * 4)]
    # many reads of one length: one register block where the launch fills whole rounds of the SIMDs (4000 reads = 2000
    # waves on 1024 SIMDs), row blocks where it does not (5000 reads) -- the library's own choice (nra_batch1d_create)
    assert int(D.padded_rows(np.full(4000, 2230), np.full(4000, 5), fold=True)[0]) == 64 * 40
    assert int(D.padded_rows(np.full(5000, 2230), np.full(5000, 5), fold=True)[0]) == 3 * 768
    # as one batch the row blocks have one height: 13 rows per lane pad 2049 + 3073 bases least (2496 + 3328 rows)
    assert D.executed_cells([("A" * 10, "CAG", "T" * 7)], [2049, 3073], [4, 4]).tolist() == [3 * 832 * (17 + 12 + 63 * 4), 4 * 832 * (17 + 12 + 63 * 4)]
    # as one batch: the lone 5-base read joins the 3-rows-per-lane bucket of the 65-base one (fold_small_buckets)
    assert D.executed_cells([("A" * 10, "CAG", "T" * 7)], [5, 65], [4, 4]).tolist() == [96 * (17 + 12 + 31 * 4)] * 2
    # a unit beyond the LDS-ring kernels keeps the DPP sweeps: full waves, 127 + 127 columns of fill
    assert D.executed_cells([("A" * 10, "ACGTACGTAC", "T" * 7)], [65], [2]).tolist() == [128 * (17 + 20 + 254)]


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


def _gpu_worker_2d(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from nanorepeat_amd import dist as D, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        j, cr, k1, k2 = _joint_case(synth)
        out = D.joint_2d_sharded(j["region"], j["reads"], cr, k1, k2, device=0)      # both ranks share GPU 0
        q.put((rank, {k: v.tolist() for k, v in out.items()}))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_joint_sharded_hip_path_two_ranks_one_gpu(capi, oracle):
    from nanorepeat_amd import synth
    world, port = 2, 35500 + os.getpid() % 2000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gpu_worker_2d, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    j, cr, k1, k2 = _joint_case(synth)
    want = oracle.joint_2d(j["region"], j["reads"], cr, k1, k2)
    for r in range(world):
        for k in ("read_strand", "best_wscore", "sum_k1", "sum_k2", "n_ties", "status"):
            assert res[r][k] == want[k].tolist(), (r, k)


def test_config4_regions_deal_evenly_over_2_4_8_ranks():
    """The strong-scaling workload north_star names (BASELINE config 4: 1000 regions x 1000 reads, mixed 3-6 bp motifs):
    the regions' costs -- executed cells, from the region descriptors alone, what bench.py --config 4 deals with -- go to
    N = 2, 4, 8 ranks with at most 3 % between the heaviest rank and the mean, every region to exactly one rank."""
    from nanorepeat_amd import dist as D, synth
    cost = np.array([synth.config4_region_cost(synth.config4_region(g), 1000) for g in range(1000)], np.int64)
    assert (cost > 0).all() and cost.max() < 8 * cost.min()          # (a region costs 0.6 - 2.7 G cells: 1000 of them deal finely)
    for world in (2, 4, 8):
        owner = D.lpt_assign(cost, world)
        assert owner.shape == cost.shape and set(owner.tolist()) == set(range(world))
        per_rank = np.bincount(owner, weights=cost, minlength=world)
        assert per_rank.max() / per_rank.mean() <= 1.03, (world, per_rank.max() / per_rank.mean())
        # reads stay with their region: a rank's shard is whole regions
        assert np.bincount(owner, minlength=world).sum() == 1000


def _nccl_worker(port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from nanorepeat_amd import dist as D, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        d = synth.make_1d(40, "TATTG", (9, 27), "ont", kwin=None, anchor=300, seed=14)
        idx = np.arange(len(d["reads"]), dtype=np.int64)
        sb = D.ShardedBatch1D(d["regions"], d["reads"], d["kmin"], d["kmax"], None, idx, len(idx), device=0)
        try:
            sb.run()
            out = sb.gather()                 # fetch + the all_gather on the nccl (= RCCL) backend
            sb.run()
            again = sb.exchange(sb.fetch_local())
        finally:
            sb.close()
        t = torch.ones(4, device="cuda")
        dist.all_reduce(t)
        q.put({k: np.asarray(v).tolist() for k, v in out.items()} | {"same": all(np.array_equal(out[k], again[k]) for k in out),
                                                                     "backend": dist.get_backend(), "sum": float(t.sum().item())})
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_sharded_batch_on_the_nccl_backend_world_size_1(capi, oracle):
    """ShardedBatch1D with its exchange on the nccl backend (RCCL), as far as one GPU allows: world size 1, device tensors,
    the same all_gather call every N takes.  (More than one rank on RCCL has not run anywhere yet: DESIGN.md 6.)"""
    from nanorepeat_amd import synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_worker, args=(37500 + os.getpid() % 2000, q))
    p.start()
    res = q.get(timeout=240)
    p.join(60)
    assert p.exitcode == 0
    d = synth.make_1d(40, "TATTG", (9, 27), "ont", kwin=None, anchor=300, seed=14)
    want = oracle.round3_1d(d["regions"], d["reads"], d["kmin"], d["kmax"])
    assert res["backend"] == "nccl" and res["same"] and res["sum"] == 4.0
    for k in ("best_score", "sum_k", "n_ties", "status"):
        assert res[k] == want[k].tolist(), k


def test_row_block_choice_agrees_with_the_measured_cases():
    """dist._prefer_row_blocks mirrors nra_batch1d_create's choice between one register block and row blocks for the reads
    of 1537 - 3072 bases; profiles/r03_row_blocks_or_one_block_54_cases.txt holds what both forms cost on an MI355X for
    54 (batch size, read length) pairs: where one form is more than 5 % faster, the model takes it."""
    import os
    import re
    from nanorepeat_amd import dist as D
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r03_row_blocks_or_one_block_54_cases.txt")
    cases = 0
    for line in open(path):
        m = re.match(r"n(\d+) k(\d+) q(\d+): one block ([\d.]+) ms .*row blocks ([\d.]+) ms", line)
        if not m:
            continue
        n, k, qmax, one, blocks = int(m.group(1)), int(m.group(2)), int(m.group(3)), float(m.group(4)), float(m.group(5))
        # the reads of a case: 90 flank bases either side of TATTG x (k .. k + 4), a few HiFi errors (tools/gpu_block_rows.py)
        q = 180 + 5 * (k + np.arange(n) % 5)
        assert abs(int(q.max()) - qmax) <= 40, (n, k, qmax)
        if q.min() <= 1536:
            continue                     # (a batch that straddles the line is another rule: the small bucket joins the blocks)
        cases += 1
        took_blocks = bool(D._prefer_row_blocks(q, np.ones(n, bool)))
        if blocks < 0.95 * one:
            assert took_blocks, (n, k, one, blocks)
        elif one < 0.95 * blocks:
            assert not took_blocks, (n, k, one, blocks)
    assert cases >= 40

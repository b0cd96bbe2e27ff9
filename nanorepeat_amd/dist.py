"""Multi-GPU sharding of the 1D path: one process per GPU, reads dealt across ranks by
estimated DP cells, no data-path collective, one small all_gather of per-read results.

The reference parallelises the same way -- regions to forked workers, nothing shared but a
result queue (nanoRepeat_bam.py:602-612, 712-728); here the unit is the read and the
"queue" is one RCCL all_gather (torch.distributed backend "nccl"; "gloo" in CPU tests).
"""
import numpy as np

from . import _capi


def estimate_cells(regions, reads, kmin, kmax, read_region=None):
    """Algorithmic DP cells of every read: qlen * sum_k (L + m*k + R)  (SURVEY.md 8d)."""
    n = len(reads)
    kmin = np.asarray(kmin, np.int64)
    kmax = np.asarray(kmax, np.int64)
    rr = np.zeros(n, np.int64) if read_region is None else np.asarray(read_region, np.int64)
    fl = np.array([len(l) + len(r) for l, _, r in regions], np.int64)[rr]
    m = np.array([len(u) for _, u, _ in regions], np.int64)[rr]
    K = np.maximum(kmax - kmin + 1, 0)
    sum_t = K * fl + m * (kmin + kmax) * K // 2
    q = np.array([len(r) for r in reads], np.int64)
    return q * sum_t


def shard_reads(cost, world):
    """Greedy longest-processing-time assignment: returns a list of index arrays, one per
    rank; deterministic, every read assigned exactly once, loads balanced to within one unit."""
    cost = np.asarray(cost, np.int64)
    order = np.argsort(-cost, kind="stable")
    load = np.zeros(world, np.int64)
    owner = np.empty(len(cost), np.int64)
    for i in order:
        r = int(np.argmin(load))
        owner[i] = r
        load[r] += cost[i]
    return [np.nonzero(owner == r)[0] for r in range(world)]


def round3_1d_sharded(regions, reads, kmin, kmax, read_region=None, sc=None, flags=0,
                      device=None, group=None, scorer=None):
    """Every rank calls this with the SAME full inputs; each scores its shard on its GPU and
    all ranks return the full per-read arrays (best_score, sum_k, n_ties, status)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    scorer = scorer or _capi.round3_1d
    kmin = np.asarray(kmin, np.int32)
    kmax = np.asarray(kmax, np.int32)
    n = len(reads)
    shards = shard_reads(estimate_cells(regions, reads, kmin, kmax, read_region), world)
    mine = shards[rank]
    rr = None if read_region is None else np.asarray(read_region, np.int32)[mine]
    if device is None:
        device = rank if not torch.cuda.is_available() else torch.cuda.current_device()
    local = scorer(regions, [reads[i] for i in mine], kmin[mine], kmax[mine], read_region=rr,
                   sc=sc, flags=flags, device=device, per_candidate=False)
    packed = np.stack([mine.astype(np.int64), local["best_score"].astype(np.int64),
                       local["sum_k"].astype(np.int64), local["n_ties"].astype(np.int64),
                       local["status"].astype(np.int64)], 1)
    out = dict(best_score=np.zeros(n, np.int32), sum_k=np.zeros(n, np.int64),
               n_ties=np.zeros(n, np.int32), status=np.zeros(n, np.uint8))
    if world == 1:
        parts = [packed]
    else:
        # one un-chunked all_gather, padded to the largest shard (rows of 5 x int64 = 40 B/read)
        backend = dist.get_backend(group)
        dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
        cap = max(len(s) for s in shards)
        buf = torch.full((cap, 5), -1, dtype=torch.int64, device=dev)
        if len(mine):
            buf[:len(mine)] = torch.from_numpy(packed).to(dev)
        gathered = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(gathered, buf, group=group)
        parts = [g.cpu().numpy()[:len(shards[r])] for r, g in enumerate(gathered)]
    for p in parts:
        idx = p[:, 0]
        out["best_score"][idx] = p[:, 1]
        out["sum_k"][idx] = p[:, 2]
        out["n_ties"][idx] = p[:, 3]
        out["status"][idx] = p[:, 4]
    return out


def estimate_cells_2d(region, reads, cell_read, cell_k1, cell_k2):
    """Algorithmic DP cells of every read of a joint batch: qlen * sum over its cells of the
    template length (every rank runs the full grid of its own reads)."""
    left, u1, mid, u2, right = region
    tl = (len(left) + len(mid) + len(right) + len(u1) * np.asarray(cell_k1, np.int64) +
          len(u2) * np.asarray(cell_k2, np.int64))
    per_read = np.bincount(np.asarray(cell_read, np.int64), weights=tl, minlength=len(reads)).astype(np.int64)
    return np.array([len(r) for r in reads], np.int64) * per_read


def joint_2d_sharded(region, reads, cell_read, cell_k1, cell_k2, read_strand=None, sc=None, flags=0,
                     device=None, group=None, scorer=None):
    """The joint grid on several GPUs: every rank calls this with the SAME full inputs (cells
    grouped by read, as nra_joint_2d wants them), scores the cells of its shard of the reads and
    all ranks return the full per-read arrays (read_strand, best_wscore, sum_k1, sum_k2, n_ties,
    status).  One all_gather of 56 B per read; per-cell arrays stay on the rank that made them."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    scorer = scorer or _capi.joint_2d
    cr = np.asarray(cell_read, np.int64)
    k1 = np.asarray(cell_k1, np.int32)
    k2 = np.asarray(cell_k2, np.int32)
    n = len(reads)
    shards = shard_reads(estimate_cells_2d(region, reads, cr, k1, k2), world)
    mine = shards[rank]
    new_index = np.full(n, -1, np.int64)
    new_index[mine] = np.arange(len(mine))
    keep = new_index[cr] >= 0                                  # cells of my reads, original (grouped) order
    order = np.argsort(new_index[cr][keep], kind="stable")      # regroup by the shard's read numbering
    st_in = None if read_strand is None else np.asarray(read_strand, np.int8)[mine]
    if device is None:
        device = rank if not torch.cuda.is_available() else torch.cuda.current_device()
    local = scorer(region, [reads[i] for i in mine], new_index[cr][keep][order].astype(np.int32), k1[keep][order],
                   k2[keep][order], read_strand=st_in, sc=sc, flags=flags, device=device)
    cols = ("read_strand", "best_wscore", "sum_k1", "sum_k2", "n_ties", "status")
    packed = np.stack([mine.astype(np.int64)] + [np.asarray(local[c]).astype(np.int64) for c in cols], 1)
    if world == 1:
        parts = [packed]
    else:
        backend = dist.get_backend(group)
        dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
        cap = max(len(s) for s in shards)
        buf = torch.full((cap, 1 + len(cols)), -1, dtype=torch.int64, device=dev)
        if len(mine):
            buf[:len(mine)] = torch.from_numpy(packed).to(dev)
        gathered = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(gathered, buf, group=group)
        parts = [g.cpu().numpy()[:len(shards[r])] for r, g in enumerate(gathered)]
    out = dict(read_strand=np.zeros(n, np.int8), best_wscore=np.zeros(n, np.int32), sum_k1=np.zeros(n, np.int64),
               sum_k2=np.zeros(n, np.int64), n_ties=np.zeros(n, np.int32), status=np.zeros(n, np.uint8))
    for p in parts:
        for j, c in enumerate(cols):
            out[c][p[:, 0]] = p[:, 1 + j]
    return out

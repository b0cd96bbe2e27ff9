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

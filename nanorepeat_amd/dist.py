"""Multi-GPU sharding of the scoring path: one process per GPU, no data-path collective, one
small all_gather of per-read results.

The reference parallelises the same way -- regions go to forked workers, nothing is shared but a
result queue (nanoRepeat_bam.py:602-612, 712-728).  Here the unit is a *region block* (a run of
one region's reads: the sweep kernels pair two reads of a region per wave, so reads of a region
stay together), blocks are dealt to ranks by the DP cells the kernels execute for them, and the
"queue" is one un-chunked RCCL all_gather of 32 B per read (torch.distributed backend "nccl";
"gloo" in the CPU tests).
"""
import heapq

import numpy as np

from . import _capi

# rows-per-lane instantiations of the sweep kernels (csrc/nra_internal.h NRA_R_LIST)
_R_LIST = np.array([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 28, 32, 40, 48], np.int64)
_CHAIN_FROM = 1536         # NRA_RING_MT_FROM: reads of more rows may run as row blocks (units the LDS ring holds; else from 3072)
_BLOCK_R = (15, 14, 13, 12)    # NRA_RING_MT_R .. NRA_RING_MT_R_MIN: rows per lane of a block, the height that pads a bucket least
_PACKED_MAX_QLEN = 6750    # doubled scores of longer reads leave the int16 range: int32 blocks, a bucket of their own
_HALF_WAVE_MAX_QLEN = 32 * 24     # NRA_RING32_MAX_R: up to here a read pair takes half a wave (k_sweep_ring32)
_RING_MAX_UNIT = 8         # NRA_SWEEP_RING_MAX_M: longer units keep the DPP sweeps (128 columns in flight)


def _rows_per_lane(need, min_reads=0, span=2):
    """Rows per lane of every read: the smallest instantiation that holds `need`, then -- like the host side of the
    library, fold_small_buckets -- a bucket of fewer than min_reads reads joins the next non-empty one within `span`
    more rows per lane (an under-filled launch costs more than the padding)."""
    idx = np.minimum(np.searchsorted(_R_LIST, need), len(_R_LIST) - 1)
    if min_reads > 0 and len(idx):
        count = np.bincount(idx, minlength=len(_R_LIST))
        target = np.arange(len(_R_LIST))
        for bi in range(len(_R_LIST) - 1):
            if count[bi] == 0 or count[bi] >= min_reads:
                continue
            for bj in range(bi + 1, len(_R_LIST)):
                if _R_LIST[bj] > _R_LIST[bi] + span:
                    break
                if count[bj] > 0:
                    count[bj] += count[bi]; count[bi] = 0
                    target[target == bi] = bj
                    break
        idx = target[idx]
    return _R_LIST[idx]


_SIMDS_DEFAULT = 1024      # MI355X: 256 compute units x 4


def _simds():
    """SIMDs of the device the batch will run on (the library's device_simds(): 4 x its compute units); MI355X's 1024 where
    no device is visible (planning on a CPU box)."""
    try:
        import torch
        if torch.cuda.is_available():
            return 4 * int(torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count)
    except Exception:
        pass
    return _SIMDS_DEFAULT


def _prefer_row_blocks(q, either):
    """The library's choice for a batch's reads of (1536, 3072] rows (nra_batch1d_create, csrc/nra_host.cpp): one
    register block per read -- one wave per SIMD, 4.3 T cells/s x the fill of the launch's last round of the SIMDs -- or
    row blocks, three waves per SIMD, 3.7 - 4.4 T cells/s from two rounds on, less below (the blocks of a read lag one another)."""
    blocks = q[(q > _CHAIN_FROM) & (q <= _PACKED_MAX_QLEN)]
    cls = q[either]
    total = {r: int(((blocks + 64 * r - 1) // (64 * r) * (64 * r)).sum()) for r in _BLOCK_R}
    best = _BLOCK_R[0]
    for r in _BLOCK_R[1:]:                            # (ties: the taller block)
        if total[r] < total[best]:
            best = r
    rows_blocks = int(((cls + 64 * best - 1) // (64 * best) * (64 * best)).sum())
    rows_single = int((64 * _R_LIST[np.searchsorted(_R_LIST, (cls + 63) // 64)]).sum())
    n = len(cls)
    simds = _simds()
    waves, nblk = (n + 1) // 2, rows_blocks / (64.0 * best) / n
    r1 = waves / simds
    e1 = r1 / np.ceil(r1)
    r2 = waves * nblk / (3.0 * simds)
    cap = 3.7 + 0.1 * (best - _BLOCK_R[-1]) + (0.4 if nblk < 2.5 else 0.0)
    return rows_single / (4.3 * e1) > rows_blocks / min(cap, 1.9 + 1.2 * r2 - 0.25 * (nblk - 2.0))


def padded_rows(qlen, unit_len=None, fold=False):
    """Rows the sweeps execute for a read of qlen bases: 32 x rows-per-lane in the half-wave kernel (reads of up to
    768 bases, units of up to 8), 64 x rows-per-lane up to 1536 bases (3072 with longer units, or where one block is the
    cheaper form for the batch), row blocks of 64 x 12 .. 15 rows beyond.  fold: the reads are one batch, whose small rows-per-lane buckets are folded as the library folds them
    and whose row blocks have one height."""
    q = np.asarray(qlen, np.int64)
    m = np.full(q.shape, 1, np.int64) if unit_len is None else np.asarray(unit_len, np.int64)
    is_chain = q > 64 * _R_LIST[-1]
    either = ~is_chain & (q > _CHAIN_FROM)            # one register block (28 .. 48 rows per lane) or row blocks
    ring_units = bool(np.all(m <= _RING_MAX_UNIT))
    blocks_chosen = ring_units and (not fold or not either.any() or bool(_prefer_row_blocks(q, either)))
    if blocks_chosen:                                 # (a read on its own, fold = False: blocks)
        is_chain = is_chain | either
    is_half = ~is_chain & (q <= _HALF_WAVE_MAX_QLEN) & (m <= _RING_MAX_UNIT)
    is_full = ~is_chain & ~is_half
    rows = np.zeros(q.shape, np.int64)
    if is_full.any():
        lanes = _rows_per_lane((q[is_full] + 63) // 64, 1024 if fold else 0)
        if fold and blocks_chosen and bool((is_chain & (q <= _PACKED_MAX_QLEN)).any()):
            # a handful of reads left in a one-block bucket above one row block joins the row blocks (fewer than 256 of a kind)
            count = np.bincount(lanes, minlength=int(_R_LIST[-1]) + 1)
            moved = (lanes > _BLOCK_R[0]) & (count[lanes] < 256)
            idx = np.nonzero(is_full)[0]
            is_chain[idx[moved]] = True
            is_full[idx[moved]] = False
            lanes = lanes[~moved]
        rows[is_full] = 64 * lanes
    wide = is_chain & (q > _PACKED_MAX_QLEN)       # int32 blocks: always 15 rows per lane
    rows[wide] = (q[wide] + 64 * _BLOCK_R[0] - 1) // (64 * _BLOCK_R[0]) * (64 * _BLOCK_R[0])
    group = is_chain & ~wide
    if group.any():
        padded = np.stack([(q[group] + 64 * r - 1) // (64 * r) * (64 * r) for r in _BLOCK_R])      # [height, read]
        if fold:                                   # one batch: one height for the bucket (ties: the taller block; _BLOCK_R descends)
            rows[group] = padded[int(np.argmin(padded.sum(axis=1)))]
        else:
            rows[group] = padded.min(axis=0)
    if is_half.any():
        rows[is_half] = 32 * _rows_per_lane((q[is_half] + 31) // 32, 2048 if fold else 0)
    return rows


def executed_cells(regions, qlen, kmax, read_region=None, fold=True):
    """DP cells the junction-decomposition kernels execute for every read (csrc/nra_host.cpp, Bucket::cells_sweep):
    padded rows x (|L| + m*kmax forward columns + |R| reverse columns + the two pipelines' fill: one column per
    lane below the first, times the skew -- m in the forward LDS-ring sweep, 1 in the reverse one; 127 + 127 in the
    DPP sweeps of long units).  This -- not the K-fold algorithmic count -- is what a shard costs.  (The kernels
    sweep the union of the windows of the two or four reads that share a wave: second-order;
    tests/test_gpu_parity.py holds the sum to 2 % of nra_stats_t's.)  fold: treat the reads as ONE batch and fold
    its small rows-per-lane buckets as the library does (a shard is folded on its own: close enough to balance by)."""
    q = np.asarray(qlen, np.int64)
    n = len(q)
    rr = np.zeros(n, np.int64) if read_region is None else np.asarray(read_region, np.int64)
    fl = np.array([len(l) + len(r) for l, _, r in regions], np.int64)[rr]
    m = np.array([len(u) for _, u, _ in regions], np.int64)[rr]
    km = np.maximum(np.asarray(kmax, np.int64), 0)
    lanes_below = np.where((q <= _HALF_WAVE_MAX_QLEN) & (m <= _RING_MAX_UNIT), 31, 63)
    fill = np.where(m <= _RING_MAX_UNIT, lanes_below * (m + 1), 254)
    return padded_rows(q, m, fold) * (fl + m * km + fill)


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


def lpt_assign(cost, world):
    """Longest-processing-time assignment of a few thousand items at most (heap)."""
    owner = np.zeros(len(cost), np.int32)
    heap = [(0, r) for r in range(world)]
    for i in np.argsort(-np.asarray(cost, np.int64), kind="stable"):
        load, r = heapq.heappop(heap)
        owner[i] = r
        heapq.heappush(heap, (load + int(cost[i]), r))
    return owner


def shard_region_blocks(cost, read_region, world, blocks_per_rank=8):
    """Owner rank of every read.  Reads of one region are cut into blocks of about
    total / (world * blocks_per_rank) cost (input order kept), and the blocks are dealt to ranks
    longest first.  Vectorised over reads; the only Python loop runs over the blocks
    (<= n_regions + world * blocks_per_rank)."""
    cost = np.asarray(cost, np.int64)
    n = len(cost)
    if n == 0 or world <= 1:
        return np.zeros(n, np.int32)
    rr = np.zeros(n, np.int64) if read_region is None else np.asarray(read_region, np.int64)
    order = np.argsort(rr, kind="stable")
    c = cost[order]
    g = rr[order]
    cs = np.cumsum(c) - c                                   # exclusive prefix over the sorted reads
    first = np.r_[True, g[1:] != g[:-1]]
    region_start = np.maximum.accumulate(np.where(first, cs, 0))
    target = max(1, int(c.sum()) // (world * blocks_per_rank))
    blk_in_region = (cs - region_start) // target
    new_blk = first | np.r_[False, blk_in_region[1:] != blk_in_region[:-1]]
    blk = np.cumsum(new_blk) - 1
    blk_owner = lpt_assign(np.bincount(blk, weights=c).astype(np.int64), world)
    owner = np.empty(n, np.int32)
    owner[order] = blk_owner[blk]
    return owner


def shard_reads(cost, world):
    """Index arrays, one per rank, for independent reads (the joint mode): costs sorted, dealt
    in snake order (0..W-1, W-1..0, ...), so the loads differ by at most the largest cost."""
    cost = np.asarray(cost, np.int64)
    order = np.argsort(-cost, kind="stable")
    pos = np.arange(len(cost)) % (2 * world)
    owner = np.empty(len(cost), np.int64)
    owner[order] = np.where(pos < world, pos, 2 * world - 1 - pos)
    return [np.nonzero(owner == r)[0] for r in range(world)]


class _FnBatch:
    """A scorer function behind the Batch interface (the CPU tests inject the oracle)."""

    def __init__(self, fn, args, kw):
        self.fn, self.args, self.kw, self.out = fn, args, kw, None

    def run(self):
        self.out = self.fn(*self.args, **self.kw)

    def sync(self):
        pass

    def fetch(self, per_candidate=False):
        return self.out

    def close(self):
        pass


class ShardedBatch1D:
    """This rank's shard of a 1D workload, resident on its GPU, plus the exchange of results.

    create: every rank passes ITS OWN reads with their global indices (`index`, positions in the
    n_total-read workload); run(): the kernels; gather(): per-read results of ALL ranks on every
    rank (one all_gather; rows of 4 x int64, padded to the largest shard).  A rank whose scorer
    fails still takes part in the collective and the error is raised on all ranks afterwards.
    """

    def __init__(self, regions, reads, kmin, kmax, read_region, index, n_total, sc=None, flags=0,
                 device=None, group=None, scorer=None):
        import torch
        import torch.distributed as dist
        self.group = group
        self.collective = dist.is_initialized()          # a one-rank group still goes through the backend
        self.world = dist.get_world_size(group) if self.collective else 1
        self.rank = dist.get_rank(group) if self.collective else 0
        self.index = np.asarray(index, np.int64)
        self.n_total = int(n_total)
        self.error = None
        self.batch = None
        if device is None:
            device = self.rank if not torch.cuda.is_available() else torch.cuda.current_device()
        try:
            if scorer is not None:
                self.batch = _FnBatch(scorer, (regions, reads, kmin, kmax),
                                      dict(read_region=read_region, sc=sc, flags=flags, device=device, per_candidate=False))
            else:
                self.batch = _capi.Batch.create_1d(regions, reads, kmin, kmax, read_region=read_region, sc=sc,
                                                   flags=flags, device=device)
        except Exception as e:          # reported to every rank by gather()
            self.error = e
        self._dev = torch.device("cpu")
        self.cap = len(self.index)
        if self.collective:
            if dist.get_backend(group) == "nccl":
                self._dev = torch.device("cuda", torch.cuda.current_device())
            sizes = torch.tensor([len(self.index)], dtype=torch.int64, device=self._dev)
            dist.all_reduce(sizes, op=dist.ReduceOp.MAX, group=group)
            self.cap = int(sizes.item())

    def run(self):
        if self.error is None:
            try:
                self.batch.run()
            except Exception as e:
                self.error = e

    def fetch_local(self):
        """Wait for this rank's kernels and fetch its per-read results (host arrays; None after an error)."""
        if self.error is None:
            try:
                self.batch.sync()
                return self.batch.fetch(per_candidate=False)
            except Exception as e:
                self.error = e
        return None

    def exchange(self, local):
        """The one exchange step: `local` (fetch_local's result) of every rank -> dict(best_score, sum_k,
        n_ties, status) for all n_total reads, on every rank.  Independent of the batch's streams, so a
        caller may run it while the next pass's kernels execute."""
        import torch
        import torch.distributed as dist
        n_local = len(self.index)
        rows = np.full((self.cap + 1, 4), -1, np.int64)
        rows[0] = (n_local, 0 if self.error is None and (local is not None or n_local == 0) else 1, 0, 0)
        if local is not None and n_local:
            rows[1:n_local + 1, 0] = self.index
            rows[1:n_local + 1, 1] = local["best_score"]
            rows[1:n_local + 1, 2] = local["sum_k"]
            rows[1:n_local + 1, 3] = (local["n_ties"].astype(np.int64) << 8) | local["status"].astype(np.int64)
        if not self.collective:
            parts = [rows]
        else:
            buf = torch.from_numpy(rows).to(self._dev)
            gathered = [torch.empty_like(buf) for _ in range(self.world)]
            dist.all_gather(gathered, buf, group=self.group)      # un-chunked: 32 B per read
            parts = [g.cpu().numpy() for g in gathered]
        failed = [r for r, p in enumerate(parts) if p[0, 1] != 0]
        if failed:
            raise RuntimeError(f"scoring failed on rank(s) {failed}" +
                               (f"; this rank: {self.error}" if self.error is not None else ""))
        out = dict(best_score=np.zeros(self.n_total, np.int32), sum_k=np.zeros(self.n_total, np.int64),
                   n_ties=np.zeros(self.n_total, np.int32), status=np.zeros(self.n_total, np.uint8))
        for p in parts:
            k = int(p[0, 0])
            idx = p[1:k + 1, 0]
            out["best_score"][idx] = p[1:k + 1, 1]
            out["sum_k"][idx] = p[1:k + 1, 2]
            out["n_ties"][idx] = p[1:k + 1, 3] >> 8
            out["status"][idx] = p[1:k + 1, 3] & 0xff
        return out

    def gather(self):
        """fetch_local + exchange: per-read results of ALL ranks on every rank."""
        return self.exchange(self.fetch_local())

    def stats(self):
        return self.batch.stats()

    def close(self):
        if self.batch is not None:
            self.batch.close()
            self.batch = None


def round3_1d_sharded(regions, reads, kmin, kmax, read_region=None, sc=None, flags=0,
                      device=None, group=None, scorer=None):
    """Every rank calls this with the SAME full inputs; each scores its region blocks on its GPU
    and all ranks return the full per-read arrays (best_score, sum_k, n_ties, status)."""
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    kmin = np.asarray(kmin, np.int32)
    kmax = np.asarray(kmax, np.int32)
    qlen = np.fromiter((len(r) for r in reads), np.int64, len(reads))
    owner = shard_region_blocks(executed_cells(regions, qlen, kmax, read_region), read_region, world)
    mine = np.nonzero(owner == rank)[0]
    rr = None if read_region is None else np.asarray(read_region, np.int32)[mine]
    sb = ShardedBatch1D(regions, [reads[i] for i in mine], kmin[mine], kmax[mine], rr, mine, len(reads),
                        sc=sc, flags=flags, device=device, group=group, scorer=scorer)
    try:
        sb.run()
        return sb.gather()
    finally:
        sb.close()


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
    cols = ("read_strand", "best_wscore", "sum_k1", "sum_k2", "n_ties", "status")
    error, local = None, None
    try:
        local = scorer(region, [reads[i] for i in mine], new_index[cr][keep][order].astype(np.int32), k1[keep][order],
                       k2[keep][order], read_strand=st_in, sc=sc, flags=flags, device=device)
    except Exception as e:              # still take part in the collective; raised on all ranks below
        error = e
    cap = max(len(s) for s in shards)
    rows = np.full((cap + 1, 1 + len(cols)), -1, np.int64)
    rows[0, :2] = (len(mine), 0 if error is None else 1)
    if local is not None and len(mine):
        rows[1:len(mine) + 1, 0] = mine
        for j, c in enumerate(cols):
            rows[1:len(mine) + 1, 1 + j] = np.asarray(local[c]).astype(np.int64)
    if world == 1:
        parts = [rows]
    else:
        backend = dist.get_backend(group)
        dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
        buf = torch.from_numpy(rows).to(dev)
        gathered = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(gathered, buf, group=group)
        parts = [g.cpu().numpy() for g in gathered]
    failed = [r for r, p in enumerate(parts) if p[0, 1] != 0]
    if failed:
        raise RuntimeError(f"joint scoring failed on rank(s) {failed}" + (f"; this rank: {error}" if error is not None else ""))
    out = dict(read_strand=np.zeros(n, np.int8), best_wscore=np.zeros(n, np.int32), sum_k1=np.zeros(n, np.int64),
               sum_k2=np.zeros(n, np.int64), n_ties=np.zeros(n, np.int32), status=np.zeros(n, np.uint8))
    for p in parts:
        k = int(p[0, 0])
        for j, c in enumerate(cols):
            out[c][p[1:k + 1, 0]] = p[1:k + 1, 1 + j]
    return out

"""ctypes binding of the C ABI declared in include/nanorepeat_amd.h.

There is no CPU path: if libnanorepeat_amd.so is missing, or no HIP device is visible,
the compute entry points raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnanorepeat_amd.so")
_LIB = None

READ_OK, READ_FALLBACK, READ_NO_RECORD, READ_SKIPPED = 0, 1, 2, 3
F_ALL_EXTENTS = 1     # explicit extents DP for every candidate (brute force)
F_TIE_EXTENTS = 2     # explicit extents DP for every top-score tie (fills cand_tstart/cand_tend)
F_BRUTE_FORCE = 4     # K independent alignments per read instead of the junction decomposition
F_TEST_CHAIN = 8      # testing only: every read in chained 128-row blocks
F_DPP_SWEEP = 16      # testing / comparison: k_sweep_pk16 instead of k_sweep_ring for unchained reads
F_NO_HALF_WAVE = 32   # testing / comparison: one read pair per wave also for reads of up to 768 bases
F_NO_JOINT_PACK = 64  # testing / comparison, 2D: int32 payload cells also outside the scoring window
F_JOINT_TAILS = 256   # testing / comparison, 2D routed grids: tail sweeps (junction at R[0]) instead of the junction at the end of mid
F_JOINT_NO_KEEP = 1024  # testing / comparison, 2D routed grids: a finer grid sweeps its reads again instead of using the column states kept from the coarse one
F_JOINT_NO_CHAIN = 512  # testing / comparison, 2D routed grids: the MID part as one systolic sweep per (read, k1) instead of column-parallel scans
F_NO_QUANTA = 2048    # testing / comparison, 1D: reverse and forward sweeps as two launches instead of one launch of quanta taken by ticket
F_QUANTA_2L = 4096    # accepted and ignored (round 4's first form of the quanta as two launches)
F_SERIAL_CHAIN = 128  # testing / comparison, 1D: a long read's row blocks one after the other in one wave

# every symbol include/nanorepeat_amd.h declares
EXPORTS = ("nra_abi_version", "nra_version", "nra_last_error", "nra_device_count",
           "nra_default_scoring", "nra_release_cached_memory", "nra_round3_1d", "nra_joint_2d", "nra_align_pairs", "nra_align_pairs_cigar", "nra_batch1d_create",
           "nra_batch2d_create", "nra_batch2d_create_reads", "nra_batch2d_set_cells", "nra_joint_grid_cells", "nra_batch2d_set_grid", "nra_batch2d_invalidate", "nra_batch2d_sweep_flanks", "nra_batch2d_refine", "nra_batch_run", "nra_batch_sync", "nra_batch_stats",
           "nra_batch1d_fetch", "nra_batch2d_fetch", "nra_batch_destroy")


E_STATE = -5      # NRA_E_STATE


class NraError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"nanorepeat_amd error {code}: {msg}")
        self.code = code


class Scoring(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("match", "mismatch", "gap_open1", "gap_ext1", "gap_open2", "gap_ext2",
                 "sc_ambi", "min_dp_score")]


class Region(C.Structure):
    _fields_ = [("left", C.c_char_p), ("unit", C.c_char_p), ("right", C.c_char_p),
                ("left_len", C.c_int32), ("unit_len", C.c_int32), ("right_len", C.c_int32)]


class JointRegion(C.Structure):
    _fields_ = [("left", C.c_char_p), ("unit1", C.c_char_p), ("mid", C.c_char_p),
                ("unit2", C.c_char_p), ("right", C.c_char_p),
                ("left_len", C.c_int32), ("unit1_len", C.c_int32), ("mid_len", C.c_int32),
                ("unit2_len", C.c_int32), ("right_len", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("n_alignments", C.c_int64), ("algorithmic_cells", C.c_int64),
                ("executed_cells", C.c_int64), ("algorithmic_bytes", C.c_int64),
                ("n_extent_tasks", C.c_int64), ("score_kernel_ms", C.c_double),
                ("extent_kernel_ms", C.c_double), ("total_ms", C.c_double),
                ("n_score_launches", C.c_int32), ("n_runs", C.c_int32),
                ("score_phase_ms", C.c_double),
                ("sum_score_kernel_ms", C.c_double), ("sum_extent_kernel_ms", C.c_double),
                ("sum_total_ms", C.c_double), ("sum_score_phase_ms", C.c_double),
                ("intermediate_bytes", C.c_int64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


def load():
    """Load the shared library (raises if it has not been built)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -m nanorepeat_amd.build` "
            "(hipcc --offload-arch=gfx950).  nanorepeat_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    p8 = C.POINTER(C.c_uint8)
    pi8 = C.POINTER(C.c_int8)
    pi32 = C.POINTER(C.c_int32)
    pi64 = C.POINTER(C.c_int64)
    vp = C.c_void_p
    lib.nra_abi_version.restype = C.c_int
    lib.nra_version.restype = C.c_char_p
    lib.nra_last_error.restype = C.c_char_p
    lib.nra_device_count.restype = C.c_int
    lib.nra_default_scoring.argtypes = [C.POINTER(Scoring)]
    lib.nra_release_cached_memory.restype = C.c_int
    lib.nra_release_cached_memory.argtypes = [C.c_int]
    lib.nra_round3_1d.restype = C.c_int
    lib.nra_round3_1d.argtypes = [C.c_int, C.POINTER(Region), C.c_int32, C.c_int32, C.c_char_p, pi64,
                                  pi32, pi32, pi32, C.POINTER(Scoring), C.c_int32,
                                  pi32, pi64, pi32, p8, pi32, pi32, pi32]
    lib.nra_joint_2d.restype = C.c_int
    lib.nra_joint_2d.argtypes = [C.c_int, C.POINTER(JointRegion), C.c_int32, C.c_char_p, pi64, pi8,
                                 C.c_int64, pi32, pi32, pi32, C.POINTER(Scoring), C.c_int32,
                                 pi32, pi32, pi32, pi64, pi64, pi32, p8]
    lib.nra_align_pairs.restype = C.c_int
    lib.nra_align_pairs.argtypes = [C.c_int, C.c_int32, C.c_char_p, pi64, C.c_int64, pi32, pi32,
                                    C.POINTER(Scoring), C.c_int32, pi32, pi32, pi32]
    lib.nra_align_pairs_cigar.restype = C.c_int
    lib.nra_align_pairs_cigar.argtypes = [C.c_int, C.c_int32, C.c_char_p, pi64, C.c_int64, pi32, pi32,
                                          C.POINTER(Scoring), C.c_int32, pi32, pi32, pi32, pi32, pi32,
                                          C.c_char_p, C.c_int64, pi64]
    lib.nra_batch1d_create.restype = C.c_int
    lib.nra_batch1d_create.argtypes = [C.c_int, C.POINTER(Region), C.c_int32, C.c_int32, C.c_char_p,
                                       pi64, pi32, pi32, pi32, C.POINTER(Scoring), C.c_int32,
                                       C.POINTER(vp)]
    lib.nra_batch2d_create.restype = C.c_int
    lib.nra_batch2d_create.argtypes = [C.c_int, C.POINTER(JointRegion), C.c_int32, C.c_char_p, pi64,
                                       pi8, C.c_int64, pi32, pi32, pi32, C.POINTER(Scoring),
                                       C.c_int32, C.POINTER(vp)]
    lib.nra_batch2d_create_reads.restype = C.c_int
    lib.nra_batch2d_create_reads.argtypes = [C.c_int, C.POINTER(JointRegion), C.c_int32, C.c_char_p, pi64,
                                             C.POINTER(Scoring), C.c_int32, C.POINTER(vp)]
    lib.nra_batch2d_set_cells.restype = C.c_int
    lib.nra_batch2d_set_cells.argtypes = [vp, pi8, C.c_int64, pi32, pi32, pi32]
    pf64 = C.POINTER(C.c_double)
    grid_args = [C.c_int32, C.c_int32, C.c_int32, pf64, pf64, C.c_int32, C.c_int32, C.c_int32, pf64, pf64]
    lib.nra_joint_grid_cells.restype = C.c_int64
    lib.nra_joint_grid_cells.argtypes = [C.c_int32] + grid_args + [C.c_int64, pi32, pi32, pi32]
    lib.nra_batch2d_set_grid.restype = C.c_int
    lib.nra_batch2d_set_grid.argtypes = [vp, pi8] + grid_args + [pi64]
    lib.nra_batch2d_sweep_flanks.restype = C.c_int
    lib.nra_batch2d_sweep_flanks.argtypes = [vp, pi8]
    lib.nra_batch2d_refine.restype = C.c_int
    lib.nra_batch2d_refine.argtypes = [vp, C.c_int32, C.c_int32, pf64, pf64, pf64, pf64]
    for f in (lib.nra_batch_run, lib.nra_batch_sync, lib.nra_batch2d_invalidate):
        f.restype = C.c_int
        f.argtypes = [vp]
    lib.nra_batch_stats.restype = C.c_int
    lib.nra_batch_stats.argtypes = [vp, C.POINTER(Stats)]
    lib.nra_batch1d_fetch.restype = C.c_int
    lib.nra_batch1d_fetch.argtypes = [vp, pi32, pi64, pi32, p8, pi32, pi32, pi32]
    lib.nra_batch2d_fetch.restype = C.c_int
    lib.nra_batch2d_fetch.argtypes = [vp, pi8, pi32, pi32, pi32, pi64, pi64, pi32, p8]
    lib.nra_batch_destroy.restype = None
    lib.nra_batch_destroy.argtypes = [vp]
    _LIB = lib
    return lib


def _check(rc):
    if rc != 0:
        raise NraError(rc, load().nra_last_error().decode(errors="replace"))


def default_scoring(**over):
    sc = Scoring()
    load().nra_default_scoring(C.byref(sc))
    for k, v in over.items():
        setattr(sc, k, v)
    return sc


def device_count():
    n = load().nra_device_count()
    if n < 0:
        raise NraError(n, load().nra_last_error().decode(errors="replace"))
    return n


def release_cached_memory(device=-1):
    """Give the library's cached device chunks and pinned buffers back to the runtime (all devices by default)."""
    _check(load().nra_release_cached_memory(device))


def _ptr(a, ty):
    return None if a is None else a.ctypes.data_as(C.POINTER(ty))


def pack_reads(reads):
    """Concatenated sequence bytes + offsets.  All-str input (the usual case) is joined once and encoded once."""
    n = len(reads)
    off = np.zeros(n + 1, dtype=np.int64)
    if n == 0:
        return b"", off
    if all(type(r) is str for r in reads):
        blob = "".join(reads)
        if blob.isascii():                      # one byte per character: lengths carry over
            np.cumsum(np.fromiter(map(len, reads), np.int64, n), out=off[1:])
            return blob.encode("ascii"), off
    bs = [r.encode() if isinstance(r, str) else bytes(r) for r in reads]
    off[1:] = np.cumsum([len(b) for b in bs])
    return b"".join(bs), off


def _regions(regions):
    arr = (Region * max(len(regions), 1))()
    keep = []
    for i, (l, u, r) in enumerate(regions):
        lb, ub, rb = l.encode(), u.encode(), r.encode()
        keep += [lb, ub, rb]
        arr[i] = Region(lb, ub, rb, len(lb), len(ub), len(rb))
    return arr, keep


def _joint_region(region):
    parts = [x.encode() for x in region]
    return JointRegion(*parts, *[len(x) for x in parts]), parts


def _outputs_1d(n, ncand):
    return dict(best_score=np.zeros(n, np.int32), sum_k=np.zeros(n, np.int64),
                n_ties=np.zeros(n, np.int32), status=np.zeros(n, np.uint8),
                cand_score=np.zeros(ncand, np.int32), cand_tstart=np.zeros(ncand, np.int32),
                cand_tend=np.zeros(ncand, np.int32))


def _outputs_2d(n, nc):
    return dict(read_strand=np.zeros(n, np.int8), cell_score=np.zeros(nc, np.int32),
                cell_wscore=np.zeros(nc, np.int32), best_wscore=np.zeros(n, np.int32),
                sum_k1=np.zeros(n, np.int64), sum_k2=np.zeros(n, np.int64),
                n_ties=np.zeros(n, np.int32), status=np.zeros(n, np.uint8))


def round3_1d(regions, reads, kmin, kmax, read_region=None, sc=None, flags=0, device=0,
              per_candidate=True):
    """One-shot nra_round3_1d.  regions = [(left, unit, right)], reads = [str]."""
    lib = load()
    sc = sc or default_scoring()
    n = len(reads)
    seqs, off = pack_reads(reads)
    kmin = np.ascontiguousarray(kmin, np.int32)
    kmax = np.ascontiguousarray(kmax, np.int32)
    rr = None if read_region is None else np.ascontiguousarray(read_region, np.int32)
    regs, keep = _regions(regions)
    ncand = int(np.maximum(kmax.astype(np.int64) - kmin + 1, 0).sum()) if per_candidate else 0
    out = _outputs_1d(n, ncand)
    pc = per_candidate
    _check(lib.nra_round3_1d(device, regs, len(regions), n, seqs, _ptr(off, C.c_int64),
                             _ptr(rr, C.c_int32), _ptr(kmin, C.c_int32), _ptr(kmax, C.c_int32),
                             C.byref(sc), flags,
                             _ptr(out["best_score"], C.c_int32), _ptr(out["sum_k"], C.c_int64),
                             _ptr(out["n_ties"], C.c_int32), _ptr(out["status"], C.c_uint8),
                             _ptr(out["cand_score"], C.c_int32) if pc else None,
                             _ptr(out["cand_tstart"], C.c_int32) if pc else None,
                             _ptr(out["cand_tend"], C.c_int32) if pc else None))
    return out


def prepared_round3_1d(regions, reads, kmin, kmax, read_region=None, sc=None, flags=0, device=0):
    """-> (call, out): `call()` is exactly one nra_round3_1d over prebuilt host buffers (ASCII reads in,
    per-read results out), so that a benchmark times the C ABI and not the Python list handling."""
    lib = load()
    sc = sc or default_scoring()
    n = len(reads)
    seqs, off = pack_reads(reads)
    kmin = np.ascontiguousarray(kmin, np.int32)
    kmax = np.ascontiguousarray(kmax, np.int32)
    rr = None if read_region is None else np.ascontiguousarray(read_region, np.int32)
    regs, keep = _regions(regions)
    out = _outputs_1d(n, 0)
    args = (device, regs, len(regions), n, seqs, _ptr(off, C.c_int64), _ptr(rr, C.c_int32),
            _ptr(kmin, C.c_int32), _ptr(kmax, C.c_int32), C.byref(sc), flags,
            _ptr(out["best_score"], C.c_int32), _ptr(out["sum_k"], C.c_int64),
            _ptr(out["n_ties"], C.c_int32), _ptr(out["status"], C.c_uint8), None, None, None)
    hold = (seqs, off, kmin, kmax, rr, regs, keep, sc)

    def call():
        _check(lib.nra_round3_1d(*args))
        return hold and out
    return call, out


def joint_2d(region, reads, cell_read, cell_k1, cell_k2, read_strand=None, sc=None, flags=0,
             device=0):
    """One-shot nra_joint_2d.  region = (left, unit1, mid, unit2, right)."""
    lib = load()
    sc = sc or default_scoring()
    n = len(reads)
    seqs, off = pack_reads(reads)
    jr, keep = _joint_region(region)
    cr = np.ascontiguousarray(cell_read, np.int32)
    k1 = np.ascontiguousarray(cell_k1, np.int32)
    k2 = np.ascontiguousarray(cell_k2, np.int32)
    out = _outputs_2d(n, len(cr))
    if read_strand is not None:
        out["read_strand"][:] = np.asarray(read_strand, np.int8)
    _check(lib.nra_joint_2d(device, C.byref(jr), n, seqs, _ptr(off, C.c_int64),
                            _ptr(out["read_strand"], C.c_int8), len(cr), _ptr(cr, C.c_int32),
                            _ptr(k1, C.c_int32), _ptr(k2, C.c_int32), C.byref(sc), flags,
                            _ptr(out["cell_score"], C.c_int32), _ptr(out["cell_wscore"], C.c_int32),
                            _ptr(out["best_wscore"], C.c_int32), _ptr(out["sum_k1"], C.c_int64),
                            _ptr(out["sum_k2"], C.c_int64), _ptr(out["n_ties"], C.c_int32),
                            _ptr(out["status"], C.c_uint8)))
    return out


class Grid:
    """One routed grid round (nra_batch2d_set_grid): per axis the grid values start + i * step, i < count, and per
    read the half-open bounds [lo, hi) (doubles) of the values it takes."""

    def __init__(self, axis1, lo1, hi1, axis2, lo2, hi2):
        self.axes = (tuple(int(x) for x in axis1), tuple(int(x) for x in axis2))      # (start, step, count)
        self.bounds = [np.ascontiguousarray(a, np.float64) for a in (lo1, hi1, lo2, hi2)]
        self.n_reads = len(self.bounds[0])
        assert all(len(a) == self.n_reads for a in self.bounds)

    def c_args(self):
        lo1, hi1, lo2, hi2 = (_ptr(a, C.c_double) for a in self.bounds)
        return (*self.axes[0], lo1, hi1, *self.axes[1], lo2, hi2)


def joint_grid_cells(grid):
    """nra_joint_grid_cells: the cells of a routed grid, (cell_read, k1, k2), on the host."""
    lib = load()
    n = lib.nra_joint_grid_cells(grid.n_reads, *grid.c_args(), 0, None, None, None)
    if n < 0:
        _check(int(n))
    cr, k1, k2 = (np.zeros(n, np.int32) for _ in range(3))
    got = lib.nra_joint_grid_cells(grid.n_reads, *grid.c_args(), n, _ptr(cr, C.c_int32), _ptr(k1, C.c_int32), _ptr(k2, C.c_int32))
    if got < 0:
        _check(int(got))
    return cr, k1, k2


def align_pairs(seqs, pair_query, pair_target, sc=None, flags=0, device=0):
    """nra_align_pairs: optimal local alignment of seqs[pair_query[i]] (query) against
    seqs[pair_target[i]] (target) -> dict(score, tstart, tend) in target coordinates."""
    lib = load()
    sc = sc or default_scoring()
    data, off = pack_reads(seqs)
    pq = np.ascontiguousarray(pair_query, np.int32)
    pt = np.ascontiguousarray(pair_target, np.int32)
    n = len(pq)
    out = dict(score=np.zeros(n, np.int32), tstart=np.zeros(n, np.int32), tend=np.zeros(n, np.int32))
    _check(lib.nra_align_pairs(device, len(seqs), data, _ptr(off, C.c_int64), n, _ptr(pq, C.c_int32),
                               _ptr(pt, C.c_int32), C.byref(sc), flags, _ptr(out["score"], C.c_int32),
                               _ptr(out["tstart"], C.c_int32), _ptr(out["tend"], C.c_int32)))
    return out


def align_pairs_cigar(seqs, pair_query, pair_target, sc=None, flags=0, device=0):
    """nra_align_pairs_cigar: like align_pairs, plus qstart/qend and the --eqx CIGAR of each pair."""
    lib = load()
    sc = sc or default_scoring()
    data, off = pack_reads(seqs)
    pq = np.ascontiguousarray(pair_query, np.int32)
    pt = np.ascontiguousarray(pair_target, np.int32)
    n = len(pq)
    out = {k: np.zeros(n, np.int32) for k in ("score", "tstart", "tend", "qstart", "qend")}
    lens = np.diff(off)
    cap = int(sum(12 * (int(lens[q]) + int(lens[t])) + 16 for q, t in zip(pq, pt))) + 16
    buf = C.create_string_buffer(cap)
    coff = np.zeros(n + 1, np.int64)
    _check(lib.nra_align_pairs_cigar(device, len(seqs), data, _ptr(off, C.c_int64), n, _ptr(pq, C.c_int32),
                                     _ptr(pt, C.c_int32), C.byref(sc), flags,
                                     *[_ptr(out[k], C.c_int32) for k in ("score", "tstart", "tend", "qstart", "qend")],
                                     buf, cap, _ptr(coff, C.c_int64)))
    raw = buf.raw
    out["cigar"] = [raw[coff[i]:coff[i + 1] - 1].decode() for i in range(n)]
    return out


TRACE_CHUNK_BYTES = 6 << 30        # nra_align_pairs_cigar keeps one trace byte per DP cell, <= 8 GiB per call


def align_pairs_cigar_chunked(seqs, pair_query, pair_target, sc=None, flags=0, device=0,
                              chunk_bytes=TRACE_CHUNK_BYTES):
    """align_pairs_cigar for any number of pairs: splits the list so that each call's trace
    (query length x target length bytes per pair) stays below `chunk_bytes`, and sends each call
    only the sequences it uses."""
    n = len(pair_query)
    out = {k: np.full(n, -1, np.int32) for k in ("score", "tstart", "tend", "qstart", "qend")}
    out["cigar"] = [""] * n
    lo = 0
    while lo < n:
        hi, used = lo, 0
        while hi < n:
            cost = len(seqs[pair_query[hi]]) * len(seqs[pair_target[hi]])
            if hi > lo and used + cost > chunk_bytes:
                break
            used += cost; hi += 1
        ids = sorted({int(x) for x in pair_query[lo:hi]} | {int(x) for x in pair_target[lo:hi]})
        local = {g: i for i, g in enumerate(ids)}
        part = align_pairs_cigar([seqs[g] for g in ids], [local[int(x)] for x in pair_query[lo:hi]],
                                 [local[int(x)] for x in pair_target[lo:hi]], sc=sc, flags=flags, device=device)
        for k in ("score", "tstart", "tend", "qstart", "qend"):
            out[k][lo:hi] = part[k]
        out["cigar"][lo:hi] = part["cigar"]
        lo = hi
    return out


class Batch:
    """Device-resident batch: create = encode + H2D, run = kernels only, fetch = D2H."""

    def __init__(self, handle, kind, n_reads, n_cand):
        self._h = handle
        self.kind = kind
        self.n_reads = n_reads
        self.n_cand = n_cand

    @classmethod
    def create_1d(cls, regions, reads, kmin, kmax, read_region=None, sc=None, flags=0, device=0):
        lib = load()
        sc = sc or default_scoring()
        seqs, off = pack_reads(reads)
        kmin = np.ascontiguousarray(kmin, np.int32)
        kmax = np.ascontiguousarray(kmax, np.int32)
        rr = None if read_region is None else np.ascontiguousarray(read_region, np.int32)
        regs, keep = _regions(regions)
        h = C.c_void_p()
        _check(lib.nra_batch1d_create(device, regs, len(regions), len(reads), seqs,
                                      _ptr(off, C.c_int64), _ptr(rr, C.c_int32),
                                      _ptr(kmin, C.c_int32), _ptr(kmax, C.c_int32), C.byref(sc),
                                      flags, C.byref(h)))
        ncand = int(np.maximum(kmax.astype(np.int64) - kmin + 1, 0).sum())
        return cls(h, 1, len(reads), ncand)

    @classmethod
    def create_2d(cls, region, reads, cell_read, cell_k1, cell_k2, read_strand=None, sc=None,
                  flags=0, device=0):
        lib = load()
        sc = sc or default_scoring()
        seqs, off = pack_reads(reads)
        jr, keep = _joint_region(region)
        cr = np.ascontiguousarray(cell_read, np.int32)
        k1 = np.ascontiguousarray(cell_k1, np.int32)
        k2 = np.ascontiguousarray(cell_k2, np.int32)
        st = None if read_strand is None else np.ascontiguousarray(read_strand, np.int8)
        h = C.c_void_p()
        _check(lib.nra_batch2d_create(device, C.byref(jr), len(reads), seqs, _ptr(off, C.c_int64),
                                      _ptr(st, C.c_int8), len(cr), _ptr(cr, C.c_int32),
                                      _ptr(k1, C.c_int32), _ptr(k2, C.c_int32), C.byref(sc), flags,
                                      C.byref(h)))
        return cls(h, 2, len(reads), len(cr))

    @classmethod
    def create_2d_reads(cls, region, reads, sc=None, flags=0, device=0):
        """The reads of a joint run, packed and resident; give it a cell list with set_cells()."""
        lib = load()
        sc = sc or default_scoring()
        seqs, off = pack_reads(reads)
        jr, keep = _joint_region(region)
        h = C.c_void_p()
        _check(lib.nra_batch2d_create_reads(device, C.byref(jr), len(reads), seqs, _ptr(off, C.c_int64),
                                            C.byref(sc), flags, C.byref(h)))
        return cls(h, 2, len(reads), 0)

    def set_cells(self, cell_read, cell_k1, cell_k2, read_strand=None):
        """The (read, k1, k2) cells of the next run (grouped by read); the previous list is dropped."""
        cr = np.ascontiguousarray(cell_read, np.int32)
        k1 = np.ascontiguousarray(cell_k1, np.int32)
        k2 = np.ascontiguousarray(cell_k2, np.int32)
        st = None if read_strand is None else np.ascontiguousarray(read_strand, np.int8)
        _check(load().nra_batch2d_set_cells(self._h, _ptr(st, C.c_int8), len(cr), _ptr(cr, C.c_int32),
                                            _ptr(k1, C.c_int32), _ptr(k2, C.c_int32)))
        self.n_cand = self._n_cells = len(cr)

    def set_grid(self, grid, read_strand=None):
        """A whole routed grid round (Grid): the library lists the cells itself.  Returns the number of cells."""
        if grid.n_reads != self.n_reads:
            raise ValueError("grid bounds must have one entry per read of the batch")
        st = None if read_strand is None else np.ascontiguousarray(read_strand, np.int8)
        n = C.c_int64(0)
        _check(load().nra_batch2d_set_grid(self._h, _ptr(st, C.c_int8), *grid.c_args(), C.byref(n)))
        self.n_cand = self._n_cells = int(n.value)
        return self.n_cand

    def sweep_flanks(self, read_strand):
        """nra_batch2d_sweep_flanks: enqueue the strand-only flank sweeps now, ahead of the cell list."""
        st = np.ascontiguousarray(read_strand, np.int8)
        if len(st) != self.n_reads:
            raise ValueError("one strand per read of the batch")
        _check(load().nra_batch2d_sweep_flanks(self._h, _ptr(st, C.c_int8)))

    def refine(self, buf1, buf2, lo1, hi1, lo2, hi2):
        """nra_batch2d_refine: the reference's round 3 enqueued behind the routed grid whose run() was just called, routed
        on the device.  False when the batch cannot (E_STATE: the caller fetches and sets the finer grid itself)."""
        b = [np.ascontiguousarray(a, np.float64) for a in (lo1, hi1, lo2, hi2)]
        if any(len(a) != self.n_reads for a in b):
            raise ValueError("refinement bounds must have one entry per read of the batch")
        rc = load().nra_batch2d_refine(self._h, int(buf1), int(buf2), *(_ptr(a, C.c_double) for a in b))
        if rc == E_STATE:
            return False
        _check(rc)
        self.n_cand = self.n_reads * 4 * int(buf1) * int(buf2)
        return True

    def invalidate(self):
        """Drop what earlier cell lists left for later ones (reverse sweeps): the next list starts like the first."""
        _check(load().nra_batch2d_invalidate(self._h))

    def run(self):
        _check(load().nra_batch_run(self._h))
        if self.kind == 2 and getattr(self, "_n_cells", None) is not None:
            self.n_cand = self._n_cells          # (a refinement of the previous run had its own per-cell layout)

    def sync(self):
        _check(load().nra_batch_sync(self._h))

    def stats(self):
        st = Stats()
        _check(load().nra_batch_stats(self._h, C.byref(st)))
        return st.as_dict()

    def fetch(self, per_candidate=True):
        lib = load()
        if self.kind == 1:
            out = _outputs_1d(self.n_reads, self.n_cand if per_candidate else 0)
            pc = per_candidate
            _check(lib.nra_batch1d_fetch(self._h, _ptr(out["best_score"], C.c_int32),
                                         _ptr(out["sum_k"], C.c_int64), _ptr(out["n_ties"], C.c_int32),
                                         _ptr(out["status"], C.c_uint8),
                                         _ptr(out["cand_score"], C.c_int32) if pc else None,
                                         _ptr(out["cand_tstart"], C.c_int32) if pc else None,
                                         _ptr(out["cand_tend"], C.c_int32) if pc else None))
            return out
        out = _outputs_2d(self.n_reads, self.n_cand if per_candidate else 0)
        pc = per_candidate
        _check(lib.nra_batch2d_fetch(self._h, _ptr(out["read_strand"], C.c_int8),
                                     _ptr(out["cell_score"], C.c_int32) if pc else None,
                                     _ptr(out["cell_wscore"], C.c_int32) if pc else None,
                                     _ptr(out["best_wscore"], C.c_int32), _ptr(out["sum_k1"], C.c_int64),
                                     _ptr(out["sum_k2"], C.c_int64), _ptr(out["n_ties"], C.c_int32),
                                     _ptr(out["status"], C.c_uint8)))
        return out

    def close(self):
        if self._h:
            load().nra_batch_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

"""ctypes loader for the CPU oracle (oracle/libnr_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under nanorepeat_amd/ may import this module.

The entry points mirror the product C ABI (include/nanorepeat_amd.h) so that a parity
test runs the same arguments through both libraries.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MODE_ORIGIN = 0
MODE_WINDOW = 1


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


def build(force=False):
    so = os.path.join(_HERE, "libnr_oracle.so")
    src = os.path.join(_HERE, "nr_oracle.c")
    srcs = [src, os.path.join(_HERE, "nr_decomp.c"), os.path.join(_HERE, "nr_oracle.h")]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(x) for x in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libnr_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def load():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = os.path.join(_HERE, "libnr_oracle.so")
    if not os.path.exists(so):
        build()
    lib = C.CDLL(so)
    p8 = C.POINTER(C.c_uint8)
    pi32 = C.POINTER(C.c_int32)
    pi64 = C.POINTER(C.c_int64)
    lib.nro_default_scoring.argtypes = [C.POINTER(Scoring)]
    lib.nro_set_threads.argtypes = [C.c_int]
    lib.nro_get_threads.restype = C.c_int
    lib.nro_encode.argtypes = [C.c_char_p, C.c_int64, p8]
    lib.nro_align.restype = C.c_int32
    lib.nro_align.argtypes = [p8, C.c_int32, p8, C.c_int32, C.POINTER(Scoring), C.c_int,
                              C.c_int32, C.c_int32, pi32, pi32]
    lib.nro_align_cigar.restype = C.c_int32
    lib.nro_align_cigar.argtypes = [p8, C.c_int32, p8, C.c_int32, C.POINTER(Scoring), C.c_int,
                                    C.c_int32, C.c_int32, C.c_char_p, C.c_int32,
                                    pi32, pi32, pi32, pi32, pi32]
    lib.nro_cigar_region_score.restype = C.c_int32
    lib.nro_cigar_region_score.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                           pi32, pi32, pi32, pi32]
    lib.nro_select_1d.restype = C.c_int
    lib.nro_select_1d.argtypes = [C.c_int32, pi32, pi32, pi32, pi32, pi32, C.c_int32, C.c_int32,
                                  pi32, pi64, pi32]
    lib.nro_round3_1d.restype = C.c_int
    lib.nro_round3_1d.argtypes = [C.POINTER(Region), C.c_int32, C.c_int32, C.c_char_p, pi64, pi32,
                                  pi32, pi32, C.POINTER(Scoring), C.c_int32,
                                  pi32, pi64, pi32, p8, pi32, pi32, pi32]
    lib.nrd_round3_1d.restype = C.c_int
    lib.nrd_round3_1d.argtypes = [C.POINTER(Region), C.c_int32, C.c_int32, C.c_char_p, pi64, pi32, pi32, pi32,
                                  C.POINTER(Scoring), pi32, pi64, pi32, p8, pi64]
    lib.nro_align_pairs.restype = C.c_int
    lib.nro_align_pairs.argtypes = [C.c_int32, C.c_char_p, pi64, C.c_int64, pi32, pi32,
                                    C.POINTER(Scoring), C.c_int32, pi32, pi32, pi32]
    lib.nro_joint_2d.restype = C.c_int
    lib.nro_joint_2d.argtypes = [C.POINTER(JointRegion), C.c_int32, C.c_char_p, pi64,
                                 C.POINTER(C.c_int8), C.c_int64, pi32, pi32, pi32,
                                 C.POINTER(Scoring), C.c_int32,
                                 pi32, pi32, pi32, pi64, pi64, pi32, p8]
    _LIB = lib
    return lib


def default_scoring(**over):
    sc = Scoring()
    load().nro_default_scoring(C.byref(sc))
    for k, v in over.items():
        setattr(sc, k, v)
    return sc


def _scoring(sc):
    """Accepts None, this module's Scoring, or any object with the same eight fields."""
    if sc is None:
        return default_scoring()
    if isinstance(sc, Scoring):
        return sc
    return Scoring(*[getattr(sc, n) for n, _ in Scoring._fields_])


def _ptr(a, ty):
    return None if a is None else a.ctypes.data_as(C.POINTER(ty))


def encode(s):
    b = s.encode() if isinstance(s, str) else bytes(s)
    out = np.empty(len(b), dtype=np.uint8)
    load().nro_encode(b, len(b), _ptr(out, C.c_uint8))
    return out


def align(q, t, sc=None, mode=MODE_ORIGIN, wa=0, wb=0):
    """(score, payload, tend) of the optimal local alignment of q vs t (str or code arrays)."""
    lib = load()
    sc = _scoring(sc)
    qc = encode(q) if not isinstance(q, np.ndarray) else np.ascontiguousarray(q, np.uint8)
    tc = encode(t) if not isinstance(t, np.ndarray) else np.ascontiguousarray(t, np.uint8)
    pay, tend = C.c_int32(0), C.c_int32(0)
    s = lib.nro_align(_ptr(qc, C.c_uint8), len(qc), _ptr(tc, C.c_uint8), len(tc), C.byref(sc),
                      mode, wa, wb, C.byref(pay), C.byref(tend))
    return s, pay.value, tend.value


def align_cigar(q, t, sc=None, mode=MODE_ORIGIN, wa=0, wb=0):
    """dict(score, cigar, tstart, tend, qstart, qend, payload) with traceback."""
    lib = load()
    sc = _scoring(sc)
    qc, tc = encode(q), encode(t)
    cap = 16 * (len(qc) + len(tc)) + 64
    buf = C.create_string_buffer(cap)
    v = [C.c_int32(0) for _ in range(5)]
    s = lib.nro_align_cigar(_ptr(qc, C.c_uint8), len(qc), _ptr(tc, C.c_uint8), len(tc),
                            C.byref(sc), mode, wa, wb, buf, cap, *[C.byref(x) for x in v])
    if s < 0:
        raise RuntimeError("cigar buffer too small")
    return dict(score=s, cigar=buf.value.decode(), tstart=v[0].value, tend=v[1].value,
                qstart=v[2].value, qend=v[3].value, payload=v[4].value)


def cigar_region_score(cigar, tstart, tend, a, b):
    """(score, num_match, num_mismatch, num_ins, num_del) -- restates tk.py:435-500."""
    v = [C.c_int32(0) for _ in range(4)]
    s = load().nro_cigar_region_score(cigar.encode(), tstart, tend, a, b, *[C.byref(x) for x in v])
    return (s,) + tuple(x.value for x in v)


def select_1d(records, left_len, right_len):
    """The oracle's 1D selector (nro_select_1d, the one nro_round3_1d applies to every read) on PAF records
    (k, AS, tstart, tend, tlen) -> (status, best_score, sum_k, n_ties).  nanoRepeat_bam.py:408-434."""
    a = np.ascontiguousarray(np.asarray(records, np.int32).reshape(-1, 5).T)
    best, sk, nt = C.c_int32(0), C.c_int64(0), C.c_int32(0)
    st = load().nro_select_1d(a.shape[1], *[_ptr(np.ascontiguousarray(a[i]), C.c_int32) for i in range(5)],
                              left_len, right_len, C.byref(best), C.byref(sk), C.byref(nt))
    return st, best.value, sk.value, nt.value


def _pack_reads(reads):
    bs = [r.encode() if isinstance(r, str) else bytes(r) for r in reads]
    off = np.zeros(len(bs) + 1, dtype=np.int64)
    if bs:
        off[1:] = np.cumsum([len(b) for b in bs])
    return b"".join(bs), off


def _regions(regions):
    arr = (Region * len(regions))()
    keep = []
    for i, (l, u, r) in enumerate(regions):
        lb, ub, rb = l.encode(), u.encode(), r.encode()
        keep += [lb, ub, rb]
        arr[i] = Region(lb, ub, rb, len(lb), len(ub), len(rb))
    return arr, keep


def round3_1d(regions, reads, kmin, kmax, read_region=None, sc=None, flags=0, threads=None, **_ignored):
    """Oracle twin of nra_round3_1d.  regions = [(left, unit, right)], reads = [str]."""
    lib = load()
    if threads is not None:
        lib.nro_set_threads(int(threads))
    sc = _scoring(sc)
    n = len(reads)
    seqs, off = _pack_reads(reads)
    kmin = np.ascontiguousarray(kmin, np.int32)
    kmax = np.ascontiguousarray(kmax, np.int32)
    rr = None if read_region is None else np.ascontiguousarray(read_region, np.int32)
    regs, keep = _regions(regions)
    ncand = int(np.maximum(kmax.astype(np.int64) - kmin + 1, 0).sum())
    out = dict(best_score=np.zeros(n, np.int32), sum_k=np.zeros(n, np.int64),
               n_ties=np.zeros(n, np.int32), status=np.zeros(n, np.uint8),
               cand_score=np.zeros(ncand, np.int32), cand_tstart=np.zeros(ncand, np.int32),
               cand_tend=np.zeros(ncand, np.int32))
    rc = lib.nro_round3_1d(regs, len(regions), n, seqs, _ptr(off, C.c_int64), _ptr(rr, C.c_int32),
                           _ptr(kmin, C.c_int32), _ptr(kmax, C.c_int32), C.byref(sc), flags,
                           _ptr(out["best_score"], C.c_int32), _ptr(out["sum_k"], C.c_int64),
                           _ptr(out["n_ties"], C.c_int32), _ptr(out["status"], C.c_uint8),
                           _ptr(out["cand_score"], C.c_int32), _ptr(out["cand_tstart"], C.c_int32),
                           _ptr(out["cand_tend"], C.c_int32))
    if rc != 0:
        raise ValueError("nro_round3_1d: bad argument")
    return out


def round3_1d_decomposed(regions, reads, kmin, kmax, read_region=None, sc=None, threads=None, **_ignored):
    """nrd_round3_1d: the junction decomposition (the HIP sweeps' algorithm) as scalar C -- NOT the oracle.  Per-read
    outputs like round3_1d plus `executed_cells`."""
    lib = load()
    if threads is not None:
        lib.nro_set_threads(int(threads))
    sc = _scoring(sc)
    n = len(reads)
    seqs, off = _pack_reads(reads)
    kmin = np.ascontiguousarray(kmin, np.int32)
    kmax = np.ascontiguousarray(kmax, np.int32)
    rr = None if read_region is None else np.ascontiguousarray(read_region, np.int32)
    regs, keep = _regions(regions)
    out = dict(best_score=np.zeros(n, np.int32), sum_k=np.zeros(n, np.int64), n_ties=np.zeros(n, np.int32),
               status=np.zeros(n, np.uint8))
    cells = C.c_int64(0)
    rc = lib.nrd_round3_1d(regs, len(regions), n, seqs, _ptr(off, C.c_int64), _ptr(rr, C.c_int32), _ptr(kmin, C.c_int32),
                           _ptr(kmax, C.c_int32), C.byref(sc), _ptr(out["best_score"], C.c_int32), _ptr(out["sum_k"], C.c_int64),
                           _ptr(out["n_ties"], C.c_int32), _ptr(out["status"], C.c_uint8), C.byref(cells))
    if rc != 0:
        raise ValueError("nrd_round3_1d: bad argument")
    out["executed_cells"] = int(cells.value)
    return out


def align_pairs(seqs, pair_query, pair_target, sc=None, flags=0, threads=None, **_ignored):
    """Oracle twin of nra_align_pairs."""
    lib = load()
    if threads is not None:
        lib.nro_set_threads(int(threads))
    sc = _scoring(sc)
    data, off = _pack_reads(seqs)
    pq = np.ascontiguousarray(pair_query, np.int32)
    pt = np.ascontiguousarray(pair_target, np.int32)
    n = len(pq)
    out = dict(score=np.zeros(n, np.int32), tstart=np.zeros(n, np.int32), tend=np.zeros(n, np.int32))
    rc = lib.nro_align_pairs(len(seqs), data, _ptr(off, C.c_int64), n, _ptr(pq, C.c_int32),
                             _ptr(pt, C.c_int32), C.byref(sc), flags, _ptr(out["score"], C.c_int32),
                             _ptr(out["tstart"], C.c_int32), _ptr(out["tend"], C.c_int32))
    if rc != 0:
        raise ValueError("nro_align_pairs: bad argument")
    return out


def align_pairs_cigar(seqs, pair_query, pair_target, sc=None, flags=0, **_ignored):
    """Oracle twin of nra_align_pairs_cigar (one nro_align_cigar per pair; small inputs only)."""
    scs = _scoring(sc)
    lo = max(1, scs.min_dp_score)
    n = len(pair_query)
    out = {k: np.full(n, -1, np.int32) for k in ("score", "tstart", "tend", "qstart", "qend")}
    out["cigar"] = [""] * n
    for i, (a, b) in enumerate(zip(pair_query, pair_target)):
        if not seqs[a] or not seqs[b]:
            continue
        r = align_cigar(seqs[a], seqs[b], sc=sc)
        if r["score"] < lo:
            continue
        for k in ("score", "tstart", "tend", "qstart", "qend"):
            out[k][i] = r[k]
        out["cigar"][i] = r["cigar"]
    return out


def joint_2d(region, reads, cell_read, cell_k1, cell_k2, read_strand=None, sc=None, flags=0,
             threads=None, **_ignored):
    """Oracle twin of nra_joint_2d.  region = (left, unit1, mid, unit2, right)."""
    lib = load()
    if threads is not None:
        lib.nro_set_threads(int(threads))
    sc = _scoring(sc)
    n = len(reads)
    seqs, off = _pack_reads(reads)
    parts = [x.encode() for x in region]
    jr = JointRegion(*parts, *[len(x) for x in parts])
    cr = np.ascontiguousarray(cell_read, np.int32)
    k1 = np.ascontiguousarray(cell_k1, np.int32)
    k2 = np.ascontiguousarray(cell_k2, np.int32)
    nc = len(cr)
    strand = np.zeros(n, np.int8) if read_strand is None else np.array(read_strand, np.int8)
    out = dict(read_strand=strand, cell_score=np.zeros(nc, np.int32),
               cell_wscore=np.zeros(nc, np.int32), best_wscore=np.zeros(n, np.int32),
               sum_k1=np.zeros(n, np.int64), sum_k2=np.zeros(n, np.int64),
               n_ties=np.zeros(n, np.int32), status=np.zeros(n, np.uint8))
    rc = lib.nro_joint_2d(C.byref(jr), n, seqs, _ptr(off, C.c_int64), _ptr(strand, C.c_int8),
                          nc, _ptr(cr, C.c_int32), _ptr(k1, C.c_int32), _ptr(k2, C.c_int32),
                          C.byref(sc), flags,
                          _ptr(out["cell_score"], C.c_int32), _ptr(out["cell_wscore"], C.c_int32),
                          _ptr(out["best_wscore"], C.c_int32), _ptr(out["sum_k1"], C.c_int64),
                          _ptr(out["sum_k2"], C.c_int64), _ptr(out["n_ties"], C.c_int32),
                          _ptr(out["status"], C.c_uint8))
    if rc != 0:
        raise ValueError("nro_joint_2d: bad argument")
    return out

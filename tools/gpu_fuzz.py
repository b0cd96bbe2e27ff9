"""Differential fuzzing of the HIP path against the CPU oracle: random regions, motifs, flank
lengths, windows, scoring parameters, N bases and adversarial reads (missing flanks, junction
indels, junk).  Usage: python tools/gpu_fuzz.py [n_rounds] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, '.')
from nanorepeat_amd import _capi as A, synth
from oracle import oracle as O

K1 = ("best_score", "sum_k", "n_ties", "status", "cand_score")
K2 = ("read_strand", "cell_score", "cell_wscore", "best_wscore", "sum_k1", "sum_k2", "n_ties", "status")


def rand_scoring(rng):
    if rng.random() < 0.6:
        return {}
    a = int(rng.integers(1, 5)); b = int(rng.integers(1, 9))
    o1 = int(rng.integers(0, 8)); e1 = int(rng.integers(1, 4))
    o2 = int(rng.integers(o1, 40)); e2 = int(rng.integers(1, e1 + 1))
    return dict(match=a, mismatch=b, gap_open1=o1, gap_ext1=e1, gap_open2=o2, gap_ext2=e2,
                sc_ambi=int(rng.integers(0, 3)), min_dp_score=int(rng.choice([0, 10, 40, 80])))


def mangle(rng, s):
    r = rng.random()
    if r < 0.15:
        i = int(rng.integers(0, max(1, len(s)))); return s[:i] + "N" * int(rng.integers(1, 4)) + s[i + 1:]
    if r < 0.25:
        return s.lower()
    return s


def fuzz_1d(rng):
    m = int(rng.integers(1, 12)); unit = synth.rand_unit(rng, m)
    L = synth.rand_seq(rng, int(rng.choice([1, 2, 9, 40, 150, 400]))); R = synth.rand_seq(rng, int(rng.choice([1, 2, 11, 60, 150, 400])))
    if rng.random() < 0.1:
        L = mangle(rng, L)
    n = int(rng.integers(1, 14)); reads, kmin, kmax = [], [], []
    for _ in range(n):
        k = int(rng.integers(0, 40)); fl, fr = int(rng.integers(0, len(L) + 1)), int(rng.integers(0, len(R) + 1))
        kind = rng.random()
        s = L[len(L) - fl:] + unit * k + R[:fr]
        if kind < 0.15: s = unit * k + R[:fr]
        elif kind < 0.3: s = L[len(L) - fl:] + unit * k
        elif kind < 0.35: s = synth.rand_seq(rng, int(rng.integers(0, 200)))
        elif kind < 0.45:
            cut = len(L[len(L) - fl:]) + m * k; d = int(rng.integers(1, 30)); s = s[:max(0, cut - d)] + s[cut + int(rng.integers(0, 30)):]
        s = synth.apply_errors(rng, s, [(0, 0, 0), "hifi", "ont_q20", "ont"][int(rng.integers(0, 4))])
        reads.append(mangle(rng, s))
        lo = max(0, k - int(rng.integers(0, 12))); hi = k + int(rng.integers(0, 12))
        if rng.random() < 0.05: lo, hi = 5, 4
        kmin.append(lo); kmax.append(hi)
    sc = rand_scoring(rng)
    o = O.round3_1d([(L, unit, R)], reads, kmin, kmax, sc=O.default_scoring(**sc))
    for flags in (0, A.F_TIE_EXTENTS, A.F_BRUTE_FORCE, A.F_TEST_CHAIN, A.F_TEST_CHAIN | A.F_SERIAL_CHAIN, A.F_DPP_SWEEP, A.F_NO_HALF_WAVE):
        if flags & A.F_TEST_CHAIN and (len(L) < 1 or len(R) < 1):
            continue
        try:
            with A.Batch.create_1d([(L, unit, R)], reads, kmin, kmax, sc=A.default_scoring(**sc), flags=flags) as b:
                b.run(); b.sync(); g = b.fetch()
        except A.NraError:
            if flags & A.F_TEST_CHAIN:      # scoring that needs the brute-force path cannot chain 128-base blocks
                continue
            raise
        keys = K1 + (("cand_tstart", "cand_tend") if flags in (A.F_TIE_EXTENTS, A.F_BRUTE_FORCE) else ())
        for k in keys:
            if not np.array_equal(g[k], o[k]):
                return dict(kind="1d", flags=flags, key=k, L=L, unit=unit, R=R, reads=reads, kmin=kmin, kmax=kmax, sc=sc,
                            got=g[k].tolist(), want=o[k].tolist())
    return None


def fuzz_1d_blocks(rng):
    """Reads of 2 - 3.6 kb: row blocks of 64 x 12 .. 15 rows (the height that pads the batch least), the one-wave chain,
    and one register block per read where that holds them (NRA_CHAIN_FROM=3072)."""
    m = int(rng.integers(1, 9)); unit = synth.rand_unit(rng, m)
    L = synth.rand_seq(rng, int(rng.choice([60, 150, 400]))); R = synth.rand_seq(rng, int(rng.choice([60, 150, 400])))
    reads, kmin, kmax = [], [], []
    for _ in range(int(rng.integers(1, 5))):
        fl, fr = int(rng.integers(20, len(L) + 1)), int(rng.integers(20, len(R) + 1))
        k = int((int(rng.integers(2060, 3600)) - fl - fr) // m)
        s = synth.apply_errors(rng, L[len(L) - fl:] + unit * k + R[:fr], ["hifi", "ont_q20", "ont"][int(rng.integers(0, 3))])
        reads.append(s if rng.random() < 0.8 else mangle(rng, s))
        lo = max(0, k - int(rng.integers(0, 4))); kmin.append(lo); kmax.append(lo + int(rng.integers(0, 6)))
    o = O.round3_1d([(L, unit, R)], reads, kmin, kmax)
    for flags, env in ((0, None), (0, "3072"), (A.F_SERIAL_CHAIN, None)):
        if env is None: os.environ.pop("NRA_CHAIN_FROM", None)
        else: os.environ["NRA_CHAIN_FROM"] = env
        try:
            g = A.round3_1d([(L, unit, R)], reads, kmin, kmax, flags=flags)
        finally:
            os.environ.pop("NRA_CHAIN_FROM", None)
        for k in K1:
            if not np.array_equal(g[k], o[k]):
                return dict(kind="1d-blocks", flags=flags, env=env, key=k, L=L, unit=unit, R=R, reads=reads, kmin=kmin, kmax=kmax,
                            got=g[k].tolist(), want=o[k].tolist())
    return None


def fuzz_1d_multi(rng):
    """Several regions in one batch (reads paired per region, buckets folded), N bases in the
    templates, skipped reads, per-candidate extents of the ties."""
    n_reg = int(rng.integers(2, 5))
    regions, reads, kmin, kmax, rr = [], [], [], [], []
    for g in range(n_reg):
        unit = synth.rand_unit(rng, int(rng.integers(1, 7)))
        L = synth.rand_seq(rng, int(rng.choice([3, 40, 200, 600]))); R = synth.rand_seq(rng, int(rng.choice([2, 50, 200, 600])))
        if rng.random() < 0.2:
            L = mangle(rng, L)
        regions.append((L, unit, R))
        for _ in range(int(rng.integers(1, 9))):
            k = int(rng.integers(0, 60))
            fl, fr = int(rng.integers(1, min(len(L), 120) + 1)), int(rng.integers(1, min(len(R), 120) + 1))
            s = synth.apply_errors(rng, L[len(L) - fl:] + unit * k + R[:fr], ["hifi", "ont_q20", "ont"][int(rng.integers(0, 3))])
            reads.append(mangle(rng, s)); rr.append(g)
            lo = max(0, k - int(rng.integers(0, 16))); hi = k + int(rng.integers(0, 16))
            if rng.random() < 0.08: lo, hi = 3, 2
            kmin.append(lo); kmax.append(hi)
    order = rng.permutation(len(reads))
    reads = [reads[i] for i in order]; kmin = [kmin[i] for i in order]; kmax = [kmax[i] for i in order]; rr = [rr[i] for i in order]
    o = O.round3_1d(regions, reads, kmin, kmax, read_region=rr)
    for flags in (0, A.F_TIE_EXTENTS, A.F_ALL_EXTENTS, A.F_TEST_CHAIN, A.F_TEST_CHAIN | A.F_SERIAL_CHAIN, A.F_TEST_CHAIN | A.F_DPP_SWEEP, A.F_NO_HALF_WAVE):
        if flags & A.F_TEST_CHAIN and any(len(L) < 1 or len(R) < 1 for L, _, R in regions):
            continue
        g = A.round3_1d(regions, reads, kmin, kmax, read_region=rr, flags=flags)
        keys, want = K1, o
        if flags == A.F_ALL_EXTENTS:
            keys = K1 + ("cand_tstart", "cand_tend")
            want = O.round3_1d(regions, reads, kmin, kmax, read_region=rr, flags=flags)
        for k in keys:
            if not np.array_equal(g[k], want[k]):
                return dict(kind="1d-multi", flags=flags, key=k, regions=regions, reads=reads, kmin=kmin, kmax=kmax, rr=rr,
                            got=g[k].tolist(), want=want[k].tolist())
        if flags == A.F_TIE_EXTENTS:      # extents are defined for the candidates tied at the best score
            ncand = np.maximum(np.array(kmax) - np.array(kmin) + 1, 0)
            tied = np.repeat(o["best_score"], ncand) == o["cand_score"]
            tied &= o["cand_score"] > 0
            for k in ("cand_tstart", "cand_tend"):
                if not np.array_equal(g[k][tied], o[k][tied]):
                    return dict(kind="1d-multi", flags=flags, key=k, regions=regions, reads=reads, kmin=kmin, kmax=kmax, rr=rr,
                                got=g[k][tied].tolist(), want=o[k][tied].tolist())
    return None


def fuzz_pairs(rng):
    """nra_align_pairs across its launch groups: short and chained queries, int32 and int64 cells (a target
    beyond 65000 columns), N bases, empty sequences."""
    seqs, pq, pt = [], [], []
    n = int(rng.integers(1, 6))
    for i in range(n):
        tl = int(rng.choice([0, 30, 400, 3000, 9000] + ([66000] if rng.random() < 0.15 else [])))
        t = synth.rand_seq(rng, tl)
        a = int(rng.integers(0, max(1, tl - 20)))
        ql = int(rng.choice([0, 25, 300, 1200] + ([3300] if rng.random() < 0.2 else [])))
        q = synth.apply_errors(rng, t[a:a + ql] if tl else synth.rand_seq(rng, ql), ["hifi", "ont"][int(rng.integers(0, 2))])
        if rng.random() < 0.3: q = synth.revcomp(q)
        seqs += [mangle(rng, t) if tl < 5000 else t, mangle(rng, q)]
        pq.append(2 * i + 1); pt.append(2 * i)
        if i and rng.random() < 0.5:
            pq.append(2 * i + 1); pt.append(2 * int(rng.integers(0, i)))
    sc = rand_scoring(rng)
    o = O.align_pairs(seqs, pq, pt, sc=O.default_scoring(**sc))
    g = A.align_pairs(seqs, pq, pt, sc=A.default_scoring(**sc))
    for k in ("score", "tstart", "tend"):
        if not np.array_equal(g[k], o[k]):
            return dict(kind="pairs", key=k, lens=[len(x) for x in seqs], pq=pq, pt=pt, sc=sc, got=g[k].tolist(), want=o[k].tolist())
    return None


def fuzz_2d(rng):
    u1 = synth.rand_unit(rng, int(rng.integers(1, 6))); u2 = synth.rand_unit(rng, int(rng.integers(1, 6)))
    # (flanks of 74 bases and more: their first |flank| - 10 >= 64 columns are swept in packed cells)
    L = synth.rand_seq(rng, int(rng.choice([1, 3, 9, 10, 11, 60, 73, 74, 75, 140, 300]))); R = synth.rand_seq(rng, int(rng.choice([1, 2, 3, 9, 10, 11, 60, 73, 74, 75, 140, 300])))
    mid = synth.rand_seq(rng, int(rng.choice([0, 1, 5, 13, 40])))
    n = int(rng.integers(1, 7)); reads, cr, k1, k2 = [], [], [], []
    for r in range(n):
        a, b = int(rng.integers(0, 25)), int(rng.integers(0, 15))
        fl, fr = int(rng.integers(0, len(L) + 1)), int(rng.integers(0, len(R) + 1))
        s = synth.apply_errors(rng, L[len(L) - fl:] + u1 * a + mid + u2 * b + R[:fr], ["hifi", "ont"][int(rng.integers(0, 2))])
        if rng.random() < 0.04:      # a read beyond one register block: scored cell by cell in chained int64 blocks
            s = synth.rand_seq(rng, int(rng.integers(1500, 2200))) + s + synth.rand_seq(rng, int(rng.integers(1500, 2200)))
        if rng.random() < 0.4: s = synth.revcomp(s)
        reads.append(mangle(rng, s))
        s1, s2 = int(rng.integers(1, 4)), int(rng.integers(1, 4))
        for x in range(max(0, a - 4), a + 4, s1):
            for y in range(max(0, b - 3), b + 3, s2):
                cr.append(r); k1.append(x); k2.append(y)
    sc = rand_scoring(rng)
    try:
        o = O.joint_2d((L, u1, mid, u2, R), reads, cr, k1, k2, sc=O.default_scoring(**sc))
    except ValueError:
        return None
    for flags in (0, A.F_BRUTE_FORCE, A.F_TEST_CHAIN, A.F_NO_JOINT_PACK):
        g = A.joint_2d((L, u1, mid, u2, R), reads, cr, k1, k2, sc=A.default_scoring(**sc), flags=flags)
        for k in K2:
            if not np.array_equal(g[k], o[k]):
                return dict(kind="2d", flags=flags, key=k, region=(L, u1, mid, u2, R), reads=reads, cr=cr, k1=k1, k2=k2, sc=sc,
                            got=g[k].tolist(), want=o[k].tolist())
    return None


def fuzz_2d_grid(rng):
    """Two routed grid rounds on one resident batch (nra_batch2d_set_grid, strands given: the strand-only kernels go
    out before the cell tasks are built) against the oracle on the cell list nra_joint_grid_cells gives."""
    u1 = synth.rand_unit(rng, int(rng.integers(1, 6))); u2 = synth.rand_unit(rng, int(rng.integers(1, 6)))
    L = synth.rand_seq(rng, int(rng.choice([1, 9, 11, 60, 74, 75, 140, 300]))); R = synth.rand_seq(rng, int(rng.choice([2, 3, 10, 60, 74, 140, 300])))
    mid = synth.rand_seq(rng, int(rng.choice([0, 1, 5, 13, 40])))
    n = int(rng.integers(1, 9)); reads, truth, strand = [], [], []
    for r in range(n):
        a, b = int(rng.integers(0, 25)), int(rng.integers(0, 15))
        fl, fr = int(rng.integers(0, len(L) + 1)), int(rng.integers(0, len(R) + 1))
        s = synth.apply_errors(rng, L[len(L) - fl:] + u1 * a + mid + u2 * b + R[:fr], ["hifi", "ont"][int(rng.integers(0, 2))])
        st = 1
        if rng.random() < 0.4: s = synth.revcomp(s); st = -1
        reads.append(mangle(rng, s)); truth.append((a, b)); strand.append(st)
    t1 = np.array([t[0] for t in truth], np.float64); t2 = np.array([t[1] for t in truth], np.float64)
    strand = np.array(strand, np.int8)
    region = (L, u1, mid, u2, R)
    flags = [0, 0, 0, A.F_JOINT_NO_CHAIN, A.F_JOINT_TAILS, A.F_JOINT_NO_KEEP][int(rng.integers(0, 6))]
    prev = None
    hist = []                                        # what the batch went through, for the report of a mismatch
    with A.Batch.create_2d_reads(region, reads, flags=flags) as b:
        if rng.random() < 0.4:                       # flank sweeps ahead of any cell list, some strands unknown
            ahead = strand.copy()
            if rng.random() < 0.4: ahead[rng.integers(0, n, size=max(1, n // 3))] = 0
            b.sweep_flanks(ahead)
            hist.append(("sweep_flanks", ahead.tolist()))
        for rnd in range(int(rng.integers(2, 5))):
            if prev is not None and rng.random() < 0.5:
                # a finer grid inside the previous bounds, the way the reference's round 3 follows round 2: within one
                # coarse step of a "size" between the read's first and last coarse value (kept column states, no sweeps)
                (p1, q1, p2, q2, s1, s2) = prev
                z1 = p1 + (q1 - p1) * rng.random(size=n); z2 = p2 + (q2 - p2) * rng.random(size=n)
                if rng.random() < 0.5: z1, z2 = np.round(z1), np.round(2 * z2) / 2
                lo1, hi1 = np.maximum(z1 - s1, p1), np.minimum(z1 + s1, q1)
                lo2, hi2 = np.maximum(z2 - s2, p2), np.minimum(z2 + s2, q2)
                a1 = (int(rng.integers(0, 3)), 1, 60); a2 = (int(rng.integers(0, 2)), 1, 40)
            else:
                a1 = (int(rng.integers(0, 6)), int(rng.integers(1, 5)), int(rng.integers(1, 12)))
                a2 = (int(rng.integers(0, 4)), int(rng.integers(1, 4)), int(rng.integers(1, 10)))
                w1, w2 = rng.integers(1, 9, size=n) + rng.choice([0.0, 0.5, 1 / 3], size=n), rng.integers(1, 6, size=n) + rng.choice([0.0, 0.5], size=n)
                lo1, hi1, lo2, hi2 = t1 - w1, t1 + w1, t2 - w2, t2 + w2
                if rng.random() < 0.3: hi2[int(rng.integers(0, n))] = -1.0          # a read without cells
            prev = (lo1, hi1, lo2, hi2, a1[1], a2[1])
            st = strand if rng.random() < 0.85 else None
            if rng.random() < 0.1: b.invalidate(); hist.append(("invalidate",))
            grid = A.Grid(a1, lo1, hi1, a2, lo2, hi2)
            hist.append(["grid", a1, a2, [x.tolist() for x in (lo1, hi1, lo2, hi2)], None if st is None else st.tolist(), False])
            cr, k1, k2 = A.joint_grid_cells(grid)
            if b.set_grid(grid, st) != len(cr):
                return dict(kind="2d-grid", key="n_cells", got=[b.n_cand], want=[len(cr)])
            if len(cr) == 0:
                continue
            b.run()
            o = O.joint_2d(region, reads, cr, k1, k2, read_strand=st)
            refined = False
            if st is not None and a1[1] > 1 and a2[1] > 1 and rng.random() < 0.6:
                # the reference's round 3 behind this grid, routed on the device (nra_batch2d_refine): the oracle on the finer
                # grid the host would have routed from this grid's results
                refined = b.refine(a1[1], a2[1], lo1, hi1, lo2, hi2)
                hist[-1][-1] = bool(refined)
                if refined:
                    ok = (o["status"] == 0) & (o["n_ties"] > 0)
                    nt = np.maximum(o["n_ties"], 1).astype(np.float64)
                    z1, z2 = o["sum_k1"] / nt, o["sum_k2"] / nt
                    f = [np.where(ok, v, 0.0) for v in (np.maximum(z1 - a1[1], lo1), np.minimum(z1 + a1[1], hi1),
                                                        np.maximum(z2 - a2[1], lo2), np.minimum(z2 + a2[1], hi2))]
                    cr, k1, k2 = A.joint_grid_cells(A.Grid((0, 1, 200), f[0], f[1], (0, 1, 100), f[2], f[3]))
                    o = O.joint_2d(region, reads, cr, k1, k2, read_strand=st)
                    o = {k: v for k, v in o.items() if not k.startswith("cell_")}          # (the refinement's per-cell arrays are laid out per read)
            b.sync(); g = b.fetch()
            has = np.zeros(n, bool); has[cr] = True
            if refined and not (np.asarray(g["status"])[~has] == 2).all():
                return dict(kind="2d-grid", key="refined status of reads without cells", flags=flags, got=np.asarray(g["status"]).tolist(), want=has.tolist())
            for k in K2:
                if k not in o: continue
                sel = has if len(o[k]) == n else slice(None)
                if not np.array_equal(np.asarray(g[k])[sel], np.asarray(o[k])[sel]):
                    return dict(kind="2d-grid", flags=flags, round=rnd, key=k, region=region, reads=reads, hist=hist, grid=(a1, a2), bounds=[x.tolist() for x in (lo1, hi1, lo2, hi2)],
                                strands=None if st is None else st.tolist(), got=np.asarray(g[k]).tolist(), want=np.asarray(o[k]).tolist())
    return None


if __name__ == "__main__":
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    t0 = time.time(); bad = 0
    hist_q = {}
    for i in range(rounds):
        # the 1D sweeps' quanta (k_sweep_ringq): the library's part size, or short parts so that these small templates are cut
        # several times -- inside the repeat, inside the reverse sweep's snapshot steps, between two flushes of the outputs
        qsteps = [None, "64", "128", "192"][int(rng.integers(0, 4))]
        hist_q[qsteps] = hist_q.get(qsteps, 0) + 1
        if qsteps is None: os.environ.pop("NRA_TEST_QSTEPS", None)
        else: os.environ["NRA_TEST_QSTEPS"] = qsteps
        for f in (fuzz_1d, fuzz_1d_multi, fuzz_1d_blocks, fuzz_2d, fuzz_2d_grid, fuzz_pairs):
            r = f(rng)
            if r is not None:
                bad += 1
                print("MISMATCH", {k: (v if k not in ("got", "want") else v[:40]) for k, v in r.items()}, flush=True)
                if bad >= 3:
                    sys.exit(1)
        if i % 20 == 19:
            print(f"round {i + 1}/{rounds} ok, {time.time() - t0:.0f} s", flush=True)
    print("fuzz done", rounds, "rounds, mismatches:", bad, "part sizes of the 1D quanta (rounds):", hist_q, flush=True)
    sys.exit(1 if bad else 0)

"""The oracle's DP against an independent formulation of the PUBLISHED objective (no GPU needed).

`oracle/nr_oracle.c` restates the aligner as ksw2's two-piece affine recurrences (five states per cell: H, E, F, E2,
F2) -- the same recurrences the HIP kernels implement in registers.  This test prices gaps the other way round: a
local alignment under a GENERAL gap cost g(l) = min(q + l*e, q2 + l*e2) (minimap2's `-O q,q2 -E e,e2`; map-ont:
min(4 + 2l, 24 + l), the two pieces cross at l = 20), O(n^3), no gap states at all:

    H(i,j) = max(0,  H(i-1,j-1) + s(i,j),  max_l H(i-l,j) - g(l),  max_l H(i,j-l) - g(l))

so an error in how the two affine pieces are opened, extended or mixed cannot cancel out.  The alignment score must
agree on every case -- long gaps on both sides of the l = 20 knee, N bases (scored -sc_ambi against anything, N
included), non-default scoring -- and the oracle's end column must be one that holds the maximum here (which of
several the oracle reports is its payload rule: the co-optimal alignment with the largest tstart).
"""
import numpy as np
import pytest


def general_gap_local(q, t, match, mismatch, go1, ge1, go2, ge2, ambi):
    """(score, end columns) of the optimal local alignments under g(l) = min(go1 + l*ge1, go2 + l*ge2): the score and the
    set of tend values (1 + column) of the cells that hold it."""
    n, m = len(q), len(t)
    if n == 0 or m == 0:
        return 0, {0}
    lmax = max(n, m)
    ls = np.arange(1, lmax + 1)
    g = np.minimum(go1 + ls * ge1, go2 + ls * ge2).astype(np.int64)
    H = np.zeros((n + 1, m + 1), np.int64)
    qa = np.frombuffer(q.encode(), np.uint8)
    ta = np.frombuffer(t.encode(), np.uint8)
    isn = lambda x: ~np.isin(x, np.frombuffer(b"ACGTacgt", np.uint8))
    sub = np.where(qa[:, None] == ta[None, :], match, -mismatch).astype(np.int64)
    sub[isn(qa), :] = -ambi
    sub[:, isn(ta)] = -ambi
    for i in range(1, n + 1):
        for j in range(1, m + 1):
            best = max(0, H[i - 1, j - 1] + sub[i - 1, j - 1])
            # a gap of l query bases ending at row i (vertical), of l template bases ending at column j (horizontal)
            v = (H[i - 1::-1, j][:i] - g[:i]).max()
            h = (H[i, j - 1::-1][:j] - g[:j]).max()
            H[i, j] = max(best, v, h)
    s = int(H.max())
    if s <= 0:
        return 0, {0}
    return s, set(np.nonzero((H == s).any(axis=0))[0].tolist())      # column index of H (1-based cell) = tend


def _cases():
    rng = np.random.default_rng(20260116)
    alpha = np.array(list("ACGT"))

    def rs(n):
        return "".join(rng.choice(alpha, n))

    def noisy(s, p):
        out = []
        for c in s:
            x = rng.random()
            if x < p:
                continue
            out.append(rng.choice(alpha) if x < 2 * p else c)
            if rng.random() < p:
                out.append(rng.choice(alpha))
        return "".join(out)

    default = (2, 4, 4, 2, 24, 1, 1)
    others = [(1, 3, 5, 2, 20, 1, 1), (2, 4, 4, 2, 24, 1, 2), (3, 5, 6, 3, 30, 1, 1), (2, 2, 2, 2, 8, 1, 1),
              (2, 4, 0, 2, 24, 1, 1)]
    cases = []
    # long gaps on either side of the knee (l = 20 for map-ont): deletions from the read and insertions into it
    for l in (1, 2, 5, 12, 18, 19, 20, 21, 22, 25, 31, 40):
        # (flanks long enough that bridging the gap beats keeping one flank alone: 2 |flank| > 24 + l)
        lo = 20 if l < 18 else 34 + l // 2
        a, b = rs(int(rng.integers(lo, lo + 8))), rs(int(rng.integers(lo, lo + 8)))
        gap = rs(l)
        cases.append((a + b, a + gap + b, default))                 # l template bases skipped
        cases.append((a + gap + b, a + b, default))                 # l read bases skipped
        cases.append((noisy(a + b, 0.04), a + gap + b, default))
        for sc in others[:2]:
            cases.append((a + b, a + gap + b, sc))
    # repeat-shaped targets (what the path aligns): L + unit^k + R against reads of another k
    for _ in range(170):
        u = str(rng.choice(["CAG", "TATTG", "AT", "GGCCCC", "A"]))
        L, R = rs(int(rng.integers(8, 24))), rs(int(rng.integers(8, 24)))
        k, kq = int(rng.integers(0, 14)), int(rng.integers(0, 14))
        t = L + u * k + R
        q = noisy(L[-int(rng.integers(4, 16)):] + u * kq + R[:int(rng.integers(4, 16))], float(rng.choice([0.0, 0.03, 0.08])))
        cases.append((q, t, default if rng.random() < 0.6 else others[int(rng.integers(0, len(others)))]))
    # N bases on either side, random pairs, empty and one-base inputs
    for _ in range(80):
        q, t = list(rs(int(rng.integers(1, 40)))), list(rs(int(rng.integers(1, 60))))
        if rng.random() < 0.5:
            core = rs(int(rng.integers(5, 25)))
            q[len(q) // 2:len(q) // 2] = core
            t[len(t) // 3:len(t) // 3] = core
        for s in (q, t):
            for _ in range(int(rng.integers(0, 4))):
                s[int(rng.integers(0, len(s)))] = "N"
        cases.append(("".join(q), "".join(t), default if rng.random() < 0.5 else others[int(rng.integers(0, len(others)))]))
    cases += [("", "ACGT", default), ("ACGT", "", default), ("A", "A", default), ("A", "C", default), ("N", "N", default),
              ("ACGTNNACGT", "ACGTNNACGT", default)]
    return cases


def test_oracle_equals_the_general_gap_cost_dp(oracle):
    cases = _cases()
    assert len(cases) >= 300
    crossed_knee = 0
    for q, t, (a, b, go1, ge1, go2, ge2, amb) in cases:
        sc = oracle.default_scoring(match=a, mismatch=b, gap_open1=go1, gap_ext1=ge1, gap_open2=go2, gap_ext2=ge2,
                                    sc_ambi=amb, min_dp_score=0)
        s, _, tend = oracle.align(q, t, sc)
        want = general_gap_local(q, t, a, b, go1, ge1, go2, ge2, amb)
        assert s == want[0] and tend in want[1], (q, t, (a, b, go1, ge1, go2, ge2, amb), (s, tend), want)
        # (a case exercises the second piece when the one-piece objective scores it lower)
        if s > 0 and (a, b, go1, ge1, go2, ge2, amb) == (2, 4, 4, 2, 24, 1, 1):
            one_piece = general_gap_local(q, t, a, b, go1, ge1, 10 ** 6, 1, amb)[0]
            crossed_knee += one_piece < s
    assert crossed_knee >= 10


def test_general_gap_cost_dp_on_hand_cases():
    """The independent DP itself on values worked out by hand."""
    d = (2, 4, 4, 2, 24, 1, 1)
    assert general_gap_local("ACGTACGTAC", "TTTACGTACGTACGGG", *d) == (20, {13})
    left, right = "ACGTTGCAAGCTTAGGCTAACGTTAGC", "TTGACCGGTATCGGATCAAGGCTTAAC"
    # 30 skipped template bases: min(4 + 60, 24 + 30) = 54
    assert general_gap_local(left + right, left + "G" * 30 + right, *d)[0] == 2 * 54 - 54
    # 10 skipped: min(4 + 20, 24 + 10) = 24
    assert general_gap_local(left + right, left + "G" * 10 + right, *d)[0] == 2 * 54 - 24
    assert general_gap_local("ACGTNACGT", "ACGTAACGT", *d)[0] == 15

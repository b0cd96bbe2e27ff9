"""The CPU oracle and the host mirror against the golden vectors captured from the
reference's own Python (tests/golden/make_golden.py).  No GPU needed."""
import numpy as np
import pytest

from nanorepeat_amd import round3 as R3, joint as J


# ------------------------------------------------------------------ pieces around the aligner
def test_cigar_region_score_matches_reference(oracle, golden_2d):
    for c in golden_2d["cigar_stats"]:
        got = oracle.cigar_region_score(c["cigar"], c["tstart"], c["tend"], c["a"], c["b"])
        want = (c["score"], c["num_match"], c["num_mismatch"], c["num_ins"], c["num_del"])
        assert got == want, c


def test_window_rule_matches_reference(golden_1d):
    for c in golden_1d["window_rule"]["cases"]:
        if c["r2"] is None:
            assert c["n_calls"] == 0
            continue
        assert R3.round3_window(c["r2"], c["fast_mode"]) == (c["kmin"], c["kmax"]), c
        assert c["bank_ok"] and c["contiguous"]
        assert c["n_templates"] == c["kmax"] - c["kmin"] + 1


def test_selector_contract_matches_reference(oracle, golden_1d):
    """The reference's 207 selector vectors (PAF records -> round-3 size, nanoRepeat_bam.py:408-434) through the ORACLE's
    selector -- nro_select_1d, the very function nro_round3_1d applies to every read -- and the host rule that turns
    (status, sum_k, n_ties) into the size (round3.py: sum_k / n_ties, the round-2 size on status 1)."""
    seen = set()
    for c in golden_1d["selector"]:
        recs = [tuple(r) for r in c["records"]]
        if not recs:
            # empty PAF: round3_estimation_from_alignment skips the read, size stays None
            assert c["result"] is None
            assert oracle.select_1d(recs, c["left_len"], c["right_len"])[0] == 2
            continue
        st, best, sk, nt = oracle.select_1d(recs, c["left_len"], c["right_len"])
        assert best == max(r[1] for r in recs)
        assert st == (0 if nt > 0 else 1)
        assert (sk / nt if st == 0 else c["r2"]) == c["result"], c
        seen.add(st)
    assert seen == {0, 1}


def test_step_size_matches_reference(golden_2d):
    for c in golden_2d["step_size"]:
        rep = J.Repeat(); rep.repeat_unit_size = c["m"]
        d = {f"r{i}": tuple(v) for i, v in enumerate(c["ranges"])}
        assert J.choose_best_step_size(rep, d) == c["step"], c


def test_template_and_anchors_match_reference(golden_2d):
    t = golden_2d["template"]
    r1 = J.Repeat().init_from_string("chr4:1200:1257:CAG:200")
    r2 = J.Repeat().init_from_string("chr4:1270:1291:CCG:20")
    name, seq = J.build_template_for_two_repeats("<L>", t["mid"], "<R>", r1, r2, 3, 2)
    assert f">{name}\n{seq}\n" == t["template_k3_k2"]


# ------------------------------------------------------------------ routing of reads to grid cells
class _Recorder:
    def __init__(self):
        self.cells = None

    def __call__(self, region, reads, cell_read, k1, k2, **kw):
        self.cells = (list(reads), list(cell_read), list(k1), list(k2))
        n = len(reads)
        return dict(status=np.full(n, 2, np.uint8), n_ties=np.zeros(n, np.int32),
                    sum_k1=np.zeros(n, np.int64), sum_k2=np.zeros(n, np.int64),
                    read_strand=np.ones(n, np.int8))


def _routing_inputs(c):
    init = J.Round1Estimation()
    init.repeat1_count_range_dict = {k: tuple(v) for k, v in c["ranges1"].items()}
    init.repeat2_count_range_dict = {k: tuple(v) for k, v in c["ranges2"].items()}
    fq = {n: f"@{n}\n{n}\n+\n!\n" for n in c["ranges1"]}       # the read "sequence" is its name
    a = J.Repeat().init_from_string("chr4:1200:1257:CAG:200")
    b = J.Repeat().init_from_string("chr4:1270:1291:CCG:20")
    return init, fq, a, b


def _cells_from_calls(calls):
    want = set()
    for tname, reads in calls:
        k1, k2 = map(int, tname.split("-"))
        for r in reads:
            want.add((r, k1, k2))
    return want


def test_grid_routing_matches_reference(golden_2d):
    chrom = "A" * 3000
    for c in golden_2d["routing"]:
        init, fq, a, b = _routing_inputs(c)
        rec = _Recorder()
        if c["round"] == 2:
            a.round1_min_size, a.round1_max_size = c["round1_min1"], c["round1_max1"]
            b.round1_min_size, b.round1_max_size = c["round1_min2"], c["round1_max2"]
            est = J.round2_estimation_of_repeat_size(init, fq, chrom, a, b, scorer=rec)
            assert (est.step_size1, est.step_size2) == (c["step1"], c["step2"])
        else:
            r2e = J.RepeatSize()
            r2e.repeat1_count_dict = {k: v[0] for k, v in c["r2sizes"].items()}
            r2e.repeat2_count_dict = {k: v[1] for k, v in c["r2sizes"].items()}
            r2e.step_size1, r2e.step_size2 = c["step"]
            J.round3_estimation_of_repeat_size(init, r2e, fq, chrom, a, b, scorer=rec)
        reads, cr, k1, k2 = rec.cells
        got = {(reads[r], x, y) for r, x, y in zip(cr, k1, k2)}
        assert got == _cells_from_calls(c["calls"])
        assert cr == sorted(cr), "cells must be grouped by read for the C ABI"


def test_grid_routing_in_the_library_equals_a_plain_enumeration(capi):
    """nra_joint_grid_cells (the routing nra_batch2d_set_grid applies) against the nested loops it replaces: grid
    values inside half-open float bounds, reads without a value on one axis dropped, k1-major order."""
    rng = np.random.default_rng(3)
    for trial in range(60):
        n = int(rng.integers(0, 40))
        a1 = (int(rng.integers(0, 30)), int(rng.integers(1, 8)), int(rng.integers(0, 25)))
        a2 = (int(rng.integers(0, 10)), int(rng.integers(1, 4)), int(rng.integers(0, 12)))
        bounds = []
        for start, step, count in (a1, a2):
            lo = rng.integers(-5, start + step * count + 5, size=n).astype(np.float64) + rng.choice([0.0, 0.5, 1 / 3, 0.25], size=n)
            hi = lo + rng.integers(-3, 40, size=n) + rng.choice([0.0, 0.5, 2 / 3], size=n)
            bounds += [lo, hi]
        got = capi.joint_grid_cells(capi.Grid(a1, bounds[0], bounds[1], a2, bounds[2], bounds[3]))
        want = []
        for r in range(n):
            g1 = [k for k in range(a1[0], a1[0] + a1[1] * a1[2], a1[1]) if bounds[0][r] <= k < bounds[1][r]]
            g2 = [k for k in range(a2[0], a2[0] + a2[1] * a2[2], a2[1]) if bounds[2][r] <= k < bounds[3][r]]
            want += [(r, x, y) for x in g1 for y in g2]
        assert list(zip(*[v.tolist() for v in got])) == want, (trial, a1, a2)
    with pytest.raises(capi.NraError):
        capi.joint_grid_cells(capi.Grid((0, 0, 3), [0.0], [1.0], (0, 1, 3), [0.0], [1.0]))


# ------------------------------------------------------------------ end to end with the oracle as scorer
def test_round3_end_to_end_matches_reference(oracle, golden_1d):
    for c in golden_1d["e2e"]:
        rr = R3.RepeatRegion()
        rr.left_anchor_seq, rr.repeat_unit_seq, rr.right_anchor_seq = c["left"], c["unit"], c["right"]
        rr.left_anchor_len, rr.right_anchor_len = len(c["left"]), len(c["right"])
        rr.chrom, rr.start_pos, rr.end_pos = "chrT", 1000, 1100
        for r in c["reads"]:
            rr.read_dict[r["name"]] = R3.Read(r["name"], r["r2"])
            rr.read_core_seq_dict[r["name"]] = r["core"]
        R3.round3_estimation("ont", c["fast_mode"], rr, 4, scorer=oracle.round3_1d)
        got = {n: (None if rd.round3_repeat_size is None else float(rd.round3_repeat_size))
               for n, rd in rr.read_dict.items()}
        assert got == c["round3"]
        assert rr.to_unique_id() == c["unique_id"]
        assert R3.output_repeat_size_1d(rr) == c["repeat_size_txt"]


def test_joint_end_to_end_matches_reference(oracle, golden_2d):
    for c in golden_2d["e2e"]:
        a = J.Repeat().init_from_string(c["repeat1"]); b = J.Repeat().init_from_string(c["repeat2"])
        a.max_size += 10; b.max_size += 10                      # nanoRepeat_joint.py:202-203
        init = J.Round1Estimation()
        fq = {}
        for r in c["reads"]:
            init.repeat1_count_range_dict[r["name"]] = tuple(r["range1"])
            init.repeat2_count_range_dict[r["name"]] = tuple(r["range2"])
            fq[r["name"]] = f"@{r['name']}\n{r['seq']}\n+\n{'!' * len(r['seq'])}\n"
        final = J.fine_tune_read_count(init, fq, c["chrom"], a, b, scorer=oracle.joint_2d)
        assert (a.round1_min_size, a.round1_max_size) == (c["round1_min1"], c["round1_max1"])
        assert (b.round1_min_size, b.round1_max_size) == (c["round1_min2"], c["round1_max2"])
        assert [final.step_size1, final.step_size2] == c["final_step"]
        assert {k: float(v) for k, v in final.repeat1_count_dict.items()} == c["k1"]
        assert {k: float(v) for k, v in final.repeat2_count_dict.items()} == c["k2"]
        _, text = J.output_repeat_size_2d("in.fastq", a.repeat_id, b.repeat_id, None,
                                          final.repeat1_count_dict, final.repeat2_count_dict)
        # the reference iterates a set() before its stable sort, so rows with equal size1 come
        # in hash order; compare header + rows as a multiset
        want, got = c["repeat_size_txt"].split("\n"), text.split("\n")
        assert want[:2] == got[:2] and sorted(want[2:]) == sorted(got[2:])
        assert [l.split("\t")[1] for l in got[2:] if l] == sorted(
            [l.split("\t")[1] for l in got[2:] if l], key=float)
        # the same reads as read groups in parallel host threads (GridSession parts): the reference's numbers again
        a2 = J.Repeat().init_from_string(c["repeat1"]); b2 = J.Repeat().init_from_string(c["repeat2"])
        a2.max_size += 10; b2.max_size += 10
        with J.GridSession(J._joint_region(c["chrom"], a2, b2), fq, scorer=oracle.joint_2d, parts=3) as sess:
            split = J.fine_tune_read_count(init, fq, c["chrom"], a2, b2, scorer=oracle.joint_2d, session=sess)
        assert [split.step_size1, split.step_size2] == c["final_step"]
        assert {k: float(v) for k, v in split.repeat1_count_dict.items()} == c["k1"]
        assert list(split.repeat1_count_dict) == list(final.repeat1_count_dict)
        assert {k: float(v) for k, v in split.repeat2_count_dict.items()} == c["k2"]


def run_wide_1d(c, scorer=None):
    """One 1D case of ref_wide.json through the host mirror (scorer: the oracle; None: the HIP library)."""
    rr = R3.RepeatRegion()
    rr.left_anchor_seq, rr.repeat_unit_seq, rr.right_anchor_seq = c["left"], c["unit"], c["right"]
    rr.left_anchor_len, rr.right_anchor_len = len(c["left"]), len(c["right"])
    rr.chrom, rr.start_pos, rr.end_pos = "chrT", 1000, 1100
    for r in c["reads"]:
        rr.read_dict[r["name"]] = R3.Read(r["name"], r["r2"])
        rr.read_core_seq_dict[r["name"]] = r["core"]
        assert list(R3.round3_window(r["r2"], c["fast_mode"])) == c["windows"][r["name"]]
    kw = {} if scorer is None else {"scorer": scorer}
    R3.round3_estimation("ont", c["fast_mode"], rr, 4, **kw)
    got = {n: (None if rd.round3_repeat_size is None else float(rd.round3_repeat_size)) for n, rd in rr.read_dict.items()}
    assert got == c["round3"]
    assert rr.to_unique_id() == c["unique_id"]
    assert R3.output_repeat_size_1d(rr) == c["repeat_size_txt"]


def run_wide_2d(c, monkeypatch, scorer=None, refine=False):
    """One joint case of ref_wide.json: the final sizes, the round-2 sizes and steps on the way, and whether round 3 ran
    (refine: round 3 behind round 2 on the device, nra_batch2d_refine -- the host then never sees the round-2 sizes)."""
    a = J.Repeat().init_from_string(c["repeat1"]); b = J.Repeat().init_from_string(c["repeat2"])
    a.max_size += 10; b.max_size += 10                      # nanoRepeat_joint.py:202-203
    init = J.Round1Estimation()
    fq = {}
    for r in c["reads"]:
        init.repeat1_count_range_dict[r["name"]] = tuple(r["range1"])
        init.repeat2_count_range_dict[r["name"]] = tuple(r["range2"])
        fq[r["name"]] = f"@{r['name']}\n{r['seq']}\n+\n{'!' * len(r['seq'])}\n"
        if refine:                  # round 1 knows the strand when it ran here; the library keeps column states only then
            init.read_strand_dict[r["name"]] = r["strand"]
    seen = {"round3_ran": False}
    r2_fn, r3_fn = J.round2_estimation_of_repeat_size, J.round3_estimation_of_repeat_size

    def spy2(*args, **kw):
        est = r2_fn(*args, **kw)
        seen["round2"] = ([est.step_size1, est.step_size2], {k: float(v) for k, v in est.repeat1_count_dict.items()},
                          {k: float(v) for k, v in est.repeat2_count_dict.items()})
        return est

    def spy3(*args, **kw):
        seen["round3_ran"] = True
        return r3_fn(*args, **kw)

    monkeypatch.setattr(J, "round2_estimation_of_repeat_size", spy2)
    monkeypatch.setattr(J, "round3_estimation_of_repeat_size", spy3)
    kw = {} if scorer is None else {"scorer": scorer}
    final = J.fine_tune_read_count(init, fq, c["chrom"], a, b, refine=refine, **kw)
    monkeypatch.undo()
    if getattr(final, "refined", False):
        assert refine and c["round3_ran"] and seen["round2"][0] == [1, 1] and not seen["round3_ran"]
    else:
        assert seen["round2"] == (c["round2"]["step"], c["round2"]["k1"], c["round2"]["k2"])
        assert seen["round3_ran"] == c["round3_ran"]
    assert [final.step_size1, final.step_size2] == c["final_step"]
    assert {k: float(v) for k, v in final.repeat1_count_dict.items()} == c["k1"]
    assert {k: float(v) for k, v in final.repeat2_count_dict.items()} == c["k2"]
    return getattr(final, "refined", False)


def test_reference_defaults_at_full_size_match_reference(oracle, golden_wide, monkeypatch):
    """ref_wide.json: the reference's default 1000-bp anchors (nanoRepeat.py:122), a read with the full K = 301 window
    (r2 >= 3000, nanoRepeat_bam.py:463-472), joint runs where round 3 ran and where a unit-step axis skipped it
    (nanoRepeat_joint.py:268) -- with the oracle answering the aligner."""
    assert [c["round3_ran"] for c in golden_wide["e2e_2d"]] == [True, False, False]
    assert max(n for c in golden_wide["e2e_1d"] for n in c["n_records"].values()) == 301
    for c in golden_wide["e2e_1d"]:
        run_wide_1d(c, oracle.round3_1d)
    for c in golden_wide["e2e_2d"]:
        run_wide_2d(c, monkeypatch, oracle.joint_2d)


def test_joint_selector_on_canned_paf(oracle, golden_2d):
    """estimate_two_repeats_from_paf on canned records: restated CIGAR rescoring + tie mean."""
    s = golden_2d["selector"]
    per_read = {}
    for line in s["lines"]:
        col = line.split("\t")
        k1, k2 = map(int, col[5].split("-"))
        cigar = [x for x in col[12:] if x.startswith("cg:Z:")][0][5:]
        tlen, ts, te = int(col[6]), int(col[7]), int(col[8])
        a = max(0, s["left_len"] - 10)
        b = min(tlen, s["left_len"] + s["m1"] * k1 + s["mid_len"] + s["m2"] * k2 + 10)
        w = oracle.cigar_region_score(cigar, ts, te, a, b)[0]
        per_read.setdefault(col[0], []).append((w, k1, k2))
    for name, recs in per_read.items():
        top = max(r[0] for r in recs)
        assert np.mean([r[1] for r in recs if r[0] == top]) == s["k1"][name]
        assert np.mean([r[2] for r in recs if r[0] == top]) == s["k2"][name]


# ------------------------------------------------------------------ oracle self-consistency
def test_window_payload_equals_rescored_traceback(oracle):
    """The DP's carried window score equals the restated reference rescoring (tk.py:435-500)
    of the oracle's own traceback CIGAR, for random noisy repeat reads."""
    import random
    rng = random.Random(3)
    from nanorepeat_amd import synth
    nrng = np.random.default_rng(9)
    n = 0
    for _ in range(150):
        L, R = synth.rand_seq(nrng, 70), synth.rand_seq(nrng, 70)
        u = rng.choice(["CAG", "TATTG", "AT", "GGCCCC"]); k = rng.randint(0, 18)
        t = L + u * k + R
        q = synth.apply_errors(nrng, L[-35:] + u * rng.randint(0, 18) + R[:35], (0.05, 0.04, 0.05))
        wa, wb = max(0, 70 - 10), min(len(t), 70 + len(u) * k + 10)
        r = oracle.align_cigar(q, t, mode=oracle.MODE_WINDOW, wa=wa, wb=wb)
        if r["score"] <= 0:
            continue
        n += 1
        assert oracle.cigar_region_score(r["cigar"], r["tstart"], r["tend"], wa, wb)[0] == r["payload"]
        s, p, te = oracle.align(q, t, mode=oracle.MODE_WINDOW, wa=wa, wb=wb)
        assert (s, p, te) == (r["score"], r["payload"], r["tend"])
        assert oracle.align(q, t)[0] == s      # same score; the end cell may differ (payload tie-break)
    assert n > 100


def test_cpu_decomposition_equals_the_oracle(oracle):
    """oracle/nr_decomp.c -- the junction decomposition of the HIP sweeps as scalar C (bench.py's same-algorithm CPU
    baseline) -- against the oracle's K independent alignments per read: exact and noisy reads (ties and ambiguous flank
    verdicts), N bases, one-base flanks, k = 0 windows, skipped and empty reads, several min_dp_score."""
    from nanorepeat_amd import synth
    rng = np.random.default_rng(11)
    statuses = set()
    for trial in range(600):
        unit = synth.rand_unit(rng, int(rng.integers(1, 7)))
        L = synth.rand_seq(rng, int(rng.choice([1, 2, 5, 12, 30, 80]))); R = synth.rand_seq(rng, int(rng.choice([1, 2, 5, 12, 30, 80])))
        reads, kmin, kmax = [], [], []
        for _ in range(int(rng.integers(1, 5))):
            kt = int(rng.integers(0, 16)); fl = int(rng.integers(0, len(L) + 1)); fr = int(rng.integers(0, len(R) + 1))
            s = L[len(L) - fl:] + unit * kt + R[:fr]
            if rng.random() < 0.6: s = synth.apply_errors(rng, s, ["hifi", "ont"][int(rng.integers(0, 2))])
            if rng.random() < 0.1 and len(s) > 2: s = s[:len(s) // 2] + "N" + s[len(s) // 2:]
            if rng.random() < 0.03: s = ""
            lo, hi = max(0, kt - int(rng.integers(0, 6))), kt + int(rng.integers(0, 6))
            if rng.random() < 0.03: lo, hi = 3, 2
            reads.append(s); kmin.append(lo); kmax.append(hi)
        sc = oracle.default_scoring(min_dp_score=int(rng.choice([0, 1, 10, 40])))
        a = oracle.round3_1d([(L, unit, R)], reads, kmin, kmax, sc=sc)
        b = oracle.round3_1d_decomposed([(L, unit, R)], reads, kmin, kmax, sc=sc)
        for k in ("best_score", "sum_k", "n_ties", "status"):
            assert np.array_equal(a[k], b[k]), (trial, k, L, unit, R, reads, kmin, kmax)
        statuses |= set(a["status"].tolist())
        assert 0 < b["executed_cells"] or all(len(r) == 0 or lo > hi for r, lo, hi in zip(reads, kmin, kmax))
    assert statuses == {0, 1, 2, 3}


def test_known_small_alignments(oracle):
    sc = oracle.default_scoring(min_dp_score=0)
    assert oracle.align("ACGTACGTAC", "TTTACGTACGTACGGG", sc) == (20, 3, 13)
    # two-piece gap: a 30-base deletion costs min(4+2*30, 24+30) = 54
    left, right = "ACGTTGCAAGCTTAGGCTAACGTTAGC" * 2, "TTGACCGGTATCGGATCAAGGCTTAAC" * 2
    gap = "G" * 30
    s, ts, te = oracle.align(left + right, left + gap + right, sc)
    assert s == 2 * len(left + right) - 54 and ts == 0 and te == len(left + gap + right)
    # N costs 1 against anything
    assert oracle.align("ACGTNACGT", "ACGTAACGT", sc)[0] == 2 * 8 - 1
    assert oracle.align("", "ACGT", sc) == (0, 0, 0)
    assert oracle.align("ACGT", "", sc) == (0, 0, 0)

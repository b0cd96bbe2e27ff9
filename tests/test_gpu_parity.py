"""Parity tests proper: the HIP path, called through the C ABI, against the CPU oracle on
the same seeded inputs (bit-exact: integer DP), the golden fixtures captured from the
reference, and size-independent properties at BASELINE config-2 size."""
import random

import numpy as np
import pytest

from nanorepeat_amd import synth, round3 as R3, joint as J

pytestmark = pytest.mark.gpu

KEYS_1D = ("best_score", "sum_k", "n_ties", "status", "cand_score", "cand_tstart", "cand_tend")
KEYS_2D = ("read_strand", "cell_score", "cell_wscore", "best_wscore", "sum_k1", "sum_k2", "n_ties", "status")


def same_1d(capi, oracle, regions, reads, kmin, kmax, read_region=None, sc_over=None, flags=0):
    kw = dict(read_region=read_region, flags=flags)
    g = capi.round3_1d(regions, reads, kmin, kmax, sc=capi.default_scoring(**(sc_over or {})), **kw)
    o = oracle.round3_1d(regions, reads, kmin, kmax, sc=oracle.default_scoring(**(sc_over or {})), **kw)
    for k in KEYS_1D:
        assert np.array_equal(g[k], o[k]), (k, np.nonzero(g[k] != o[k])[0][:8], g[k][:8], o[k][:8])
    return g


def same_2d(capi, oracle, region, reads, cr, k1, k2, strand=None, sc_over=None):
    g = capi.joint_2d(region, reads, cr, k1, k2, read_strand=strand, sc=capi.default_scoring(**(sc_over or {})))
    o = oracle.joint_2d(region, reads, cr, k1, k2, read_strand=strand, sc=oracle.default_scoring(**(sc_over or {})))
    for k in KEYS_2D:
        assert np.array_equal(g[k], o[k]), (k, np.nonzero(g[k] != o[k])[0][:8], g[k][:8], o[k][:8])
    return g


# ------------------------------------------------------------------ 1D
@pytest.mark.parametrize("unit,alleles,model,anchor,flank", [
    ("TATTG", (8, 30), "ont", 300, 100),
    ("CAG", (5, 41), "ont_q20", 250, 80),
    ("AT", (0, 17), "hifi", 120, 60),
    ("GGCCCC", (3, 12), "ont", 1000, 100),
    ("A", (10, 25), "ont", 200, 70),
])
def test_1d_random_matches_oracle(capi, oracle, unit, alleles, model, anchor, flank):
    d = synth.make_1d(20, unit, alleles, model, kwin=(0, max(alleles) + 12), anchor=anchor, flank=flank,
                      seed=anchor + len(unit))
    g = same_1d(capi, oracle, d["regions"], d["reads"], d["kmin"], d["kmax"])
    ok = g["status"] == 0
    assert ok.mean() > 0.8
    est = g["sum_k"][ok] / g["n_ties"][ok]
    assert np.mean(np.abs(est - d["k_true"][ok]) <= (2 if len(unit) == 1 else 1)) > 0.75


def test_1d_reference_window_rule_and_all_extents(capi, oracle):
    d = synth.make_1d(16, "TATTG", (12, 33), "ont", kwin=None, anchor=300, seed=11)
    same_1d(capi, oracle, d["regions"], d["reads"], d["kmin"], d["kmax"])
    same_1d(capi, oracle, d["regions"], d["reads"], d["kmin"], d["kmax"], flags=capi.F_ALL_EXTENTS)
    d2 = synth.make_1d(8, "TATTG", (12, 33), "ont", kwin=None, anchor=300, seed=12, fast_mode=True)
    same_1d(capi, oracle, d2["regions"], d2["reads"], d2["kmin"], d2["kmax"])


def test_1d_edge_cases(capi, oracle):
    rng = np.random.default_rng(5)
    L, R, u = synth.rand_seq(rng, 200), synth.rand_seq(rng, 200), "TATTG"
    core = lambda k, fl=80: L[-fl:] + u * k + R[:fl]
    reads = [
        "",                                   # empty read
        core(7),                              # skipped (kmin > kmax)
        synth.rand_seq(rng, 120),             # junk: nothing reaches min_dp_score -> NO_RECORD
        core(0),                              # true k = 0 (template L+R is a legal candidate)
        core(9),                              # error-free
        u * 9 + R[:80],                       # no left flank: alignment starts inside the repeat -> flank test fails
        L[-80:] + u * 9,                      # no right flank
        core(6).lower(),                      # lower case
        core(6).replace("T", "U", 3),         # U == T
        core(8)[:40] + "N" * 3 + core(8)[43:],  # N in the read
        "ACGT",                               # shorter than anything useful
        core(5, 200),                         # whole flanks
    ]
    kmin = np.array([0, 5, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0], np.int32)
    kmax = np.array([9, 4, 12, 6, 20, 20, 20, 15, 15, 15, 5, 12], np.int32)
    g = same_1d(capi, oracle, [(L, u, R)], reads, kmin, kmax)
    assert g["status"].tolist()[:7] == [2, 3, 2, 0, 0, 1, 1]
    assert g["sum_k"][3] == 0 and g["n_ties"][3] == 1
    assert g["sum_k"][4] == 9 * g["n_ties"][4]
    same_1d(capi, oracle, [(L, u, R)], reads, kmin, kmax, flags=capi.F_ALL_EXTENTS)
    # N inside the flanks / unit as well
    Ln = L[:150] + "NN" + L[152:]
    same_1d(capi, oracle, [(Ln, "TANTG", R)], reads, kmin, kmax)


def test_1d_ties_average(capi, oracle):
    """A read whose repeat tract lost part of one unit ties between neighbouring k."""
    rng = np.random.default_rng(8)
    L, R, u = synth.rand_seq(rng, 150), synth.rand_seq(rng, 150), "CAG"
    reads = []
    for k in (6, 10, 14):
        s = L[-70:] + u * k + R[:70]
        reads += [s, s[:70 + 3 * 3] + s[70 + 3 * 3 + 1:], s[:70 + 6] + "T" + s[70 + 6:]]
    g = same_1d(capi, oracle, [(L, u, R)], reads, [0] * 9, [22] * 9, sc_over=dict(min_dp_score=40))
    assert (g["n_ties"] >= 1).all()


def test_1d_many_regions_and_buckets(capi, oracle):
    """Several regions with different motif lengths in one batch; read lengths spanning many
    rows-per-lane instantiations, up to the 3072-row limit."""
    rng = np.random.default_rng(21)
    regions, reads, rr, kmin, kmax = [], [], [], [], []
    for g, (m, fl) in enumerate([(3, 60), (4, 80), (5, 100), (6, 50), (2, 30)]):
        unit = synth.rand_unit(rng, m)
        L, R = synth.rand_seq(rng, 90 + 10 * g), synth.rand_seq(rng, 110 - 10 * g)
        regions.append((L, unit, R))
        for k in (1, 7, 20, 45, 90):
            s = synth.apply_errors(rng, L[-min(fl, len(L)):] + unit * k + R[:min(fl, len(R))], "ont_q20")
            reads.append(s); rr.append(g); kmin.append(max(0, k - 2)); kmax.append(k + 2)
    # long reads: q ~ 700, 1300, 2100, 3072 (R = 11, 22, 40, 48; anything longer is chained)
    unit = regions[2][1]
    L, R = regions[2][0], regions[2][2]
    for q in (700, 1300, 2100, 3072):
        k = (q - 160) // 5
        s = (L[-80:] + unit * k + R[:80])
        s = synth.apply_errors(rng, s, "hifi")[:q]
        reads.append(s); rr.append(2); kmin.append(k - 1); kmax.append(k + 1)
    same_1d(capi, oracle, regions, reads, kmin, kmax, read_region=rr)


def test_1d_scoring_variants(capi, oracle):
    d = synth.make_1d(8, "CAG", (7, 19), "ont", kwin=(0, 26), anchor=200, seed=31)
    for over in (dict(min_dp_score=0), dict(match=1, mismatch=3, gap_open1=5, gap_ext1=2, gap_open2=20, gap_ext2=1),
                 dict(match=3, mismatch=2, gap_open1=2, gap_ext1=3, gap_open2=9, gap_ext2=2, sc_ambi=2, min_dp_score=30)):
        same_1d(capi, oracle, d["regions"], d["reads"], d["kmin"], d["kmax"], sc_over=over)


def test_1d_large_match_scores(capi, oracle):
    """Scores that outgrow the packed int16 sweeps (doubled scores: match x length > 8000) move to the
    chained int32 sweeps; a scoring scheme the sweeps cannot hold (table bytes) falls back to the packed
    brute-force kernel, which refuses scores beyond 24000."""
    d = synth.make_1d(6, "TATTG", (20, 60), "ont_q20", kwin=(10, 70), anchor=600, flank=500, seed=77)   # reads of 1.1-1.3 kb
    assert max(len(r) for r in d["reads"]) > 1000
    for over in (dict(match=7, mismatch=9, gap_open1=9, gap_ext1=4, gap_open2=40, gap_ext2=2),          # 7 x 1300 > 8000
                 dict(match=6, mismatch=12, gap_open1=12, gap_ext1=6, gap_open2=72, gap_ext2=3, min_dp_score=240)):
        same_1d(capi, oracle, d["regions"], d["reads"], d["kmin"], d["kmax"], sc_over=over)
    with pytest.raises(capi.NraError) as e:
        capi.round3_1d(d["regions"], d["reads"], d["kmin"], d["kmax"], sc=capi.default_scoring(match=30, mismatch=40, gap_open1=40, gap_ext1=20, gap_open2=240, gap_ext2=10))
    assert e.value.code == -3


def test_1d_errors(capi):
    L, R = "ACGT" * 20, "TTGCA" * 16
    with pytest.raises(capi.NraError) as e:
        capi.round3_1d([(L, "CAG", R)], ["A" * 200001], [0], [1])       # NRA_MAX_QLEN
    assert e.value.code == -3
    with pytest.raises(capi.NraError) as e:                              # long reads need the sweeps
        capi.round3_1d([(L, "CAG", R)], ["A" * 8001], [0], [1], flags=capi.F_BRUTE_FORCE)
    assert e.value.code == -3
    out = capi.round3_1d([(L, "CAG", R)], ["A" * 8001], [0], [1])        # ... which take them
    assert out["status"].tolist() == [2]
    with pytest.raises(capi.NraError) as e:
        capi.round3_1d([(L, "", R)], ["ACGT"], [0], [1])
    assert e.value.code == -1
    with pytest.raises(capi.NraError) as e:
        capi.round3_1d([(L, "CAG", R)], ["ACGT"], [0], [3000000])        # NRA_MAX_TLEN_WIDE
    assert e.value.code == -3
    out = capi.round3_1d([(L, "CAG", R)], [], [], [])
    assert len(out["status"]) == 0


def test_1d_batch_rerun_is_idempotent(capi):
    d = synth.make_1d(12, "TATTG", (9, 22), "ont", kwin=(0, 30), anchor=300, seed=41)
    with capi.Batch.create_1d(d["regions"], d["reads"], d["kmin"], d["kmax"]) as b:
        b.run(); b.sync(); first = b.fetch()
        b.run(); b.run(); b.sync(); again = b.fetch()
        st = b.stats()
    for k in KEYS_1D:
        assert np.array_equal(first[k], again[k]), k
    assert st["n_alignments"] == 12 * 31 and st["score_kernel_ms"] > 0
    assert st["algorithmic_cells"] > st["executed_cells"] > 0     # the decomposition shares work across k


def test_round3_estimation_golden_end_to_end(capi, golden_1d):
    """Reference round3_estimation results (fixture) reproduced by the product host path + HIP."""
    for c in golden_1d["e2e"]:
        rr = R3.RepeatRegion()
        rr.left_anchor_seq, rr.repeat_unit_seq, rr.right_anchor_seq = c["left"], c["unit"], c["right"]
        rr.chrom, rr.start_pos, rr.end_pos = "chrT", 1000, 1100
        for r in c["reads"]:
            rr.read_dict[r["name"]] = R3.Read(r["name"], r["r2"])
            rr.read_core_seq_dict[r["name"]] = r["core"]
        R3.round3_estimation("ont", c["fast_mode"], rr, 4)
        got = {n: (None if rd.round3_repeat_size is None else float(rd.round3_repeat_size))
               for n, rd in rr.read_dict.items()}
        assert got == c["round3"]
        assert R3.output_repeat_size_1d(rr) == c["repeat_size_txt"]


def test_the_forms_of_the_ring_sweeps_agree(capi, oracle, monkeypatch):
    """One launch of quanta taken by ticket (k_sweep_ringq, the default: parts of 512 steps), the same with parts of 64, 128
    and 320 -> 256 steps (NRA_TEST_QSTEPS: cuts inside the repeat, inside the snapshot steps of the reverse sweep, between
    two flushes of the outputs, parts that hold no boundary step at all) and the reverse / forward launches of round 3
    (NRA_F_NO_QUANTA): the same per-read and per-candidate results, equal to the oracle on a sample -- templates whose
    first boundary lies before the first cut (short L, kmin 0), reads of both kernels (half-wave and full-wave), N bases."""
    rng = np.random.default_rng(33)
    cases = [synth.config2(n_reads=600)]
    left, right = synth.rand_seq(rng, 40), synth.rand_seq(rng, 70)                 # first boundary at column 39: no plain part
    reads = [synth.apply_errors(rng, left[-30:] + "CAG" * k + right[:50], "ont") for k in rng.integers(0, 40, size=64)]
    reads[5] = reads[5][:20] + "N" + reads[5][20:]
    cases.append(dict(regions=[(left, "CAG", right)], reads=reads, kmin=np.zeros(64, np.int32), kmax=np.full(64, 45, np.int32)))
    for d in cases:
        res = {}
        for name, fl, qsteps in (("tickets", 0, None), ("parts of 64", 0, "64"), ("parts of 128", 0, "128"), ("parts of 256", 0, "320"),
                                 ("reverse / forward", capi.F_NO_QUANTA, None)):
            if qsteps is None:
                monkeypatch.delenv("NRA_TEST_QSTEPS", raising=False)
            else:
                monkeypatch.setenv("NRA_TEST_QSTEPS", qsteps)
            with capi.Batch.create_1d(d["regions"], d["reads"], d["kmin"], d["kmax"], flags=fl) as b:
                b.run(); b.sync()
                res[name] = b.fetch()
                b.run(); b.sync()
                again = b.fetch()
                for k in again:
                    assert np.array_equal(again[k], res[name][k]), (name, "second run", k)
        monkeypatch.delenv("NRA_TEST_QSTEPS", raising=False)
        for name in res:
            for k in res["tickets"]:
                assert np.array_equal(res[name][k], res["tickets"][k]), (name, k)
        o = oracle.round3_1d(d["regions"], d["reads"][:48], d["kmin"][:48], d["kmax"][:48])
        for k in ("best_score", "sum_k", "n_ties", "status"):
            assert np.array_equal(res["tickets"][k][:48], o[k]), k


def test_regions_of_a_bed_in_one_call_equal_region_by_region(capi, oracle):
    """INTEGRATION.md's primary 1D stub: the regions a worker would take one after the other (nanoRepeat_bam.py:602-612) in
    ONE round3_estimation_regions call -- 15 regions of mixed motifs x 50 reads, reference window rule, a read without a
    round-2 size, a junk read -- against 15 single-region calls (the per-region drop-in, :446-450) and the oracle."""
    rng = np.random.default_rng(15)
    regions = []
    for g in range(15):
        unit = synth.rand_unit(rng, int(rng.integers(2, 7)))
        left, right = synth.rand_seq(rng, 400), synth.rand_seq(rng, 400)
        rr = R3.RepeatRegion()
        rr.left_anchor_seq, rr.repeat_unit_seq, rr.right_anchor_seq = left, unit, right
        rr.chrom, rr.start_pos, rr.end_pos = "chr1", 1000 * g, 1000 * g + 50
        alleles = (int(rng.integers(5, 60)), int(rng.integers(5, 60)))
        for i in range(50):
            kt = alleles[i % 2]
            core = synth.apply_errors(rng, left[-100:] + unit * kt + right[:100], "ont_q20")
            r2 = max(0.0, kt + float(rng.normal(0, 1)))
            if i == 7:
                r2 = None                                  # no round-2 estimate: skipped (:460)
            if i == 9:
                core = synth.rand_seq(rng, 80)             # nothing reaches min_dp_score: no record (:421)
            rr.read_dict[f"g{g}r{i}"] = R3.Read(f"g{g}r{i}", r2)
            rr.read_core_seq_dict[f"g{g}r{i}"] = core
        regions.append(rr)

    def sizes():
        return [[None if rd.round3_repeat_size is None else float(rd.round3_repeat_size) for rd in rr.read_dict.values()] for rr in regions]

    def reset():
        for rr in regions:
            for rd in rr.read_dict.values():
                rd.round3_repeat_size = None

    R3.round3_estimation_regions("ont", False, regions, 4)                          # one call
    one_call = sizes()
    reset()
    for rr in regions:
        R3.round3_estimation("ont", False, rr, 4)                                   # region by region
    by_region = sizes()
    reset()
    R3.round3_estimation_regions("ont", False, regions, 4, scorer=oracle.round3_1d)
    want = sizes()
    assert one_call == by_region == want
    assert sum(v is None for reg in want for v in reg) >= 30                        # the skipped and the junk read of every region


# ------------------------------------------------------------------ 2D
def _cells(j, step=2):
    cr, k1, k2 = [], [], []
    for r in range(len(j["reads"])):
        for a in range(int(j["range1"][r][0]), int(j["range1"][r][1]), step):
            for b in range(int(j["range2"][r][0]), int(j["range2"][r][1]), step):
                cr.append(r); k1.append(a); k2.append(b)
    return cr, k1, k2


def test_2d_random_matches_oracle(capi, oracle):
    j = synth.make_joint(10, alleles=((6, 4), (11, 3)), read_len=500, read_sd=30, anchor=300, seed=3)
    cr, k1, k2 = _cells(j)
    g = same_2d(capi, oracle, j["region"], j["reads"], cr, k1, k2)
    assert np.array_equal(g["read_strand"], j["strand"])
    same_2d(capi, oracle, j["region"], j["reads"], cr, k1, k2, strand=j["strand"])
    same_2d(capi, oracle, j["region"], j["reads"], cr, k1, k2, strand=-j["strand"])     # forced wrong strand


def test_2d_htt_like_and_edges(capi, oracle):
    j = synth.make_joint(6, alleles=((17, 10), (30, 7)), read_len=700, read_sd=40, anchor=1000, seed=7)
    cr, k1, k2 = _cells(j, step=3)
    same_2d(capi, oracle, j["region"], j["reads"], cr, k1, k2)
    # short flanks: the window is clipped at both template ends (left_len < 10, wb > tlen)
    left, u1, mid, u2, right = j["region"]
    region = (left[-6:], u1, "", u2, right[:5])
    reads = [u1 * 9 + u2 * 6, synth.revcomp(u1 * 9 + u2 * 6), "ACGTNNACGT" + u1 * 5 + "N" + u2 * 4, ""]
    cr2, a2, b2 = [], [], []
    for r in (0, 1, 2):
        for a in range(3, 12, 2):
            for b in range(2, 9, 2):
                cr2.append(r); a2.append(a); b2.append(b)
    same_2d(capi, oracle, region, reads, cr2, a2, b2, sc_over=dict(min_dp_score=20))
    # read without cells, zero cells at all
    g = capi.joint_2d(j["region"], j["reads"][:2], [], [], [])
    assert g["status"].tolist() == [2, 2]


def test_joint_golden_end_to_end(capi, golden_2d):
    for c in golden_2d["e2e"]:
        a = J.Repeat().init_from_string(c["repeat1"]); b = J.Repeat().init_from_string(c["repeat2"])
        a.max_size += 10; b.max_size += 10
        init = J.Round1Estimation()
        fq = {}
        for r in c["reads"]:
            init.repeat1_count_range_dict[r["name"]] = tuple(r["range1"])
            init.repeat2_count_range_dict[r["name"]] = tuple(r["range2"])
            fq[r["name"]] = f"@{r['name']}\n{r['seq']}\n+\n{'!' * len(r['seq'])}\n"
        final = J.fine_tune_read_count(init, fq, c["chrom"], a, b)
        assert [final.step_size1, final.step_size2] == c["final_step"]
        assert {k: float(v) for k, v in final.repeat1_count_dict.items()} == c["k1"]
        assert {k: float(v) for k, v in final.repeat2_count_dict.items()} == c["k2"]


def test_reference_defaults_at_full_size_golden_end_to_end(capi, golden_wide, monkeypatch):
    """ref_wide.json through the product host path + HIP: 1000-bp anchors, the K = 301 window of a 6.2 kb core, joint runs
    with and without round 3 (a unit-step axis), the round-2 sizes on the way."""
    from test_oracle_golden import run_wide_1d, run_wide_2d
    for c in golden_wide["e2e_1d"]:
        run_wide_1d(c)
    refined = 0
    for c in golden_wide["e2e_2d"]:
        run_wide_2d(c, monkeypatch)                       # two grid calls, round 3 routed on the host
        refined += bool(run_wide_2d(c, monkeypatch, refine=True))      # round 3 behind round 2 on the device
    assert refined == sum(c["round3_ran"] for c in golden_wide["e2e_2d"])


# ------------------------------------------------------------------ full size: properties
@pytest.fixture(scope="module")
def config2_run(capi):
    d = synth.config2()
    with capi.Batch.create_1d(d["regions"], d["reads"], d["kmin"], d["kmax"], flags=capi.F_TIE_EXTENTS) as b:
        b.run(); b.sync()
        return d, b.fetch(), b.stats()


def test_config2_full_size_properties(capi, oracle, config2_run):
    d, g, st = config2_run
    n = len(d["reads"])
    assert st["n_alignments"] == n * 196
    # every read has a record, the size lies inside its window, scores are bounded by 2*qlen
    assert (g["status"] <= 1).all() and (g["status"] == 0).mean() > 0.98
    ok = g["status"] == 0
    est = g["sum_k"][ok] / g["n_ties"][ok]
    assert (est >= 5).all() and (est <= 200).all()
    qlen = np.array([len(r) for r in d["reads"]])
    assert (g["best_score"] <= 2 * qlen).all() and (g["best_score"] > qlen).all()
    assert np.mean(np.abs(est - d["k_true"][ok]) <= 1) > 0.95
    # Score(k) of a read is maximal at the reported ties and the profile is unimodal around them
    sc = g["cand_score"].reshape(n, 196)
    assert np.array_equal(sc.max(1), g["best_score"])
    for r in range(0, n, 997):
        kbest = int(np.argmax(sc[r]))
        assert (np.diff(sc[r][:max(kbest - 3, 1)]) >= -2).all()
    # extents were produced exactly for the top-score ties
    ties = sc == g["best_score"][:, None]
    assert np.array_equal(g["cand_tstart"].reshape(n, 196) >= 0, ties)
    # a seeded sample of reads -- 128 of either allele -- against the oracle at full problem size
    rng = np.random.default_rng(1)
    pick = np.concatenate([rng.choice(np.nonzero(d["k_true"] == a)[0], 128, replace=False) for a in (40, 150)])
    o = oracle.round3_1d(d["regions"], [d["reads"][i] for i in pick], d["kmin"][pick], d["kmax"][pick])
    for k in ("best_score", "sum_k", "n_ties", "status"):
        assert np.array_equal(g[k][pick], o[k]), k
    assert np.array_equal(sc[pick].ravel(), o["cand_score"])


def test_long_cores_chained_int32_sweeps_equal_oracle(capi, oracle):
    """Cores beyond the packed int16 sweeps (3072 rows / doubled scores of 8000): 10 kb, 16 kb and 20 kb
    expansions of a 5 bp motif run as chained row blocks in int32 cells, one read per wave.  The 16 kb
    read gets the reference's full window for r2 = 3000 (buffer capped at 150: K = 301,
    nanoRepeat_bam.py:463-472); the others a narrow window (the oracle is K full DPs of 10^8 cells)."""
    from nanorepeat_amd.round3 import round3_window
    rng = np.random.default_rng(77)
    L, R = synth.rand_seq(rng, 1000), synth.rand_seq(rng, 1000)
    reads, kmin, kmax = [], [], []
    for k_true, win in ((2000, (1997, 2003)), (3000, round3_window(3000.4, False)), (4000, (3998, 4001)),
                        (30, (20, 40)), (700, (690, 712))):
        reads.append(synth.apply_errors(rng, L[-100:] + "TATTG" * k_true + R[:100], "ont_q20"))
        kmin.append(win[0]); kmax.append(win[1])
    assert (kmin[1], kmax[1]) == (2850, 3150) and len(reads[2]) > 19000
    o = oracle.round3_1d([(L, "TATTG", R)], reads, kmin, kmax)
    for flags in (0, capi.F_TIE_EXTENTS, capi.F_SERIAL_CHAIN):
        g = capi.round3_1d([(L, "TATTG", R)], reads, kmin, kmax, flags=flags)
        for k in ("best_score", "sum_k", "n_ties", "status", "cand_score"):
            assert np.array_equal(g[k], o[k]), (flags, k, g[k][:8], o[k][:8])
        if flags & capi.F_TIE_EXTENTS:
            ties = o["cand_tstart"] >= 0
            assert np.array_equal(g["cand_tstart"][ties], o["cand_tstart"][ties])
            assert np.array_equal(g["cand_tend"][ties], o["cand_tend"][ties])
    assert (g["status"] == 0).all() and g["best_score"][2] > 32767


def test_large_row_counts_park_the_junction_in_agprs(capi, oracle):
    """Reads of 1.7-3.0 kb take 28-48 rows per lane in the packed sweeps: the forward sweep then keeps the R
    side of the junction in accumulation registers.  Two reads per bucket (pairs) plus a single, N bases."""
    rng = np.random.default_rng(91)
    L, R = synth.rand_seq(rng, 400), synth.rand_seq(rng, 400)
    reads, kmin, kmax = [], [], []
    for k_true in (300, 310, 350, 360, 440, 450, 520, 530, 560):          # cores of 1.7 ... 3.0 kb
        s = synth.apply_errors(rng, L[-100:] + "TATTG" * k_true + R[:100], "hifi")
        if k_true == 440:
            s = s[:700] + "NN" + s[702:]
        reads.append(s[:3072]); kmin.append(k_true - 6); kmax.append(k_true + 6)
    assert max(len(r) for r in reads) > 2900
    o = oracle.round3_1d([(L, "TATTG", R)], reads, kmin, kmax)
    for flags in (0, capi.F_TIE_EXTENTS, capi.F_DPP_SWEEP):
        g = capi.round3_1d([(L, "TATTG", R)], reads, kmin, kmax, flags=flags)
        for k in ("best_score", "sum_k", "n_ties", "status", "cand_score"):
            assert np.array_equal(g[k], o[k]), (flags, k, g[k][:8], o[k][:8])


def test_joint_resident_reads_across_cell_lists(capi, oracle):
    """nra_batch2d_create_reads + set_cells: several cell lists on one resident batch (the two grid rounds).
    The reverse sweeps of a read are made once per strand and reused by later lists; every list's results
    equal the one-shot call's and the oracle's, also after a strand changes or is left to the probe."""
    j = synth.make_joint(10, alleles=((9, 5), (14, 3)), read_len=520, read_sd=30, anchor=300, seed=73)
    n = len(j["reads"])

    def cells(step, shift):
        cr, k1, k2 = [], [], []
        for r in range(n):
            if (r + shift) % 4 == 3:
                continue                                   # some reads sit a list out
            for a in range(max(0, int(j["truth"][r][0]) - 4 + shift), int(j["truth"][r][0]) + 5, step):
                for b in range(max(0, int(j["truth"][r][1]) - 2), int(j["truth"][r][1]) + 3, step):
                    cr.append(r); k1.append(a); k2.append(b)
        return cr, k1, k2

    true_strand = j["strand"].astype(np.int8)
    flipped = true_strand.copy(); flipped[[1, 4]] *= -1
    with capi.Batch.create_2d_reads(j["region"], j["reads"]) as b:
        for step, shift, strands in ((2, 0, true_strand), (1, 1, true_strand), (1, 0, flipped), (1, 2, None),
                                     (2, 1, true_strand)):
            cr, k1, k2 = cells(step, shift)
            b.set_cells(cr, k1, k2, strands)
            b.run(); b.sync()
            g = b.fetch()
            o = oracle.joint_2d(j["region"], j["reads"], cr, k1, k2, read_strand=strands)
            one = capi.joint_2d(j["region"], j["reads"], cr, k1, k2, read_strand=strands)
            has = np.zeros(n, bool); has[cr] = True
            for key in o:
                per_read = len(o[key]) == n
                sel = has if per_read else slice(None)
                assert np.array_equal(np.asarray(g[key])[sel], np.asarray(o[key])[sel]), (step, shift, key)
                assert np.array_equal(np.asarray(one[key])[sel], np.asarray(o[key])[sel]), (step, shift, key)


def test_joint_routed_grid_equals_its_cell_list(capi, oracle):
    """nra_batch2d_set_grid (the library routes reads to grid cells and builds the sweeps from the per-read rows)
    against nra_batch2d_set_cells on the cell list nra_joint_grid_cells gives for the same grid, and the oracle:
    coarse and fine grids, fractional bounds (round 3 compares means), reads without cells, several rounds on one
    resident batch, strands given and probed."""
    j = synth.make_joint(14, alleles=((9, 5), (14, 3)), read_len=520, read_sd=30, anchor=300, seed=74)
    n = len(j["reads"])
    t1, t2 = j["truth"][:, 0].astype(np.float64), j["truth"][:, 1].astype(np.float64)
    strands = j["strand"].astype(np.int8)
    rounds = [((2, 3, 7), t1 - 5, t1 + 6, (0, 2, 6), t2 - 3, t2 + 4, strands),
              ((5, 1, 14), t1 - 1.5, t1 + 1.5, (1, 1, 8), t2 - 1 / 3, t2 + 2.5, strands),
              ((0, 4, 6), t1 - 8, t1 + 3, (0, 1, 9), np.where(np.arange(n) % 5 == 2, 99.0, t2 - 2), t2 + 2, None),
              ((3, 1, 20), t1 - 2, t1 + 2.5, (2, 1, 2), t2 - 9, t2 + 9, strands)]
    # the three forms a routed grid can take: junction at the end of mid with a read's MID sweeps chained in one wave (the
    # default), with one MID sweep per (read, k1), and the tail sweeps with the junction at R[0]
    with capi.Batch.create_2d_reads(j["region"], j["reads"]) as by_grid, \
            capi.Batch.create_2d_reads(j["region"], j["reads"], flags=capi.F_JOINT_NO_CHAIN) as by_grid_mid, \
            capi.Batch.create_2d_reads(j["region"], j["reads"], flags=capi.F_JOINT_TAILS) as by_grid_tails, \
            capi.Batch.create_2d_reads(j["region"], j["reads"]) as by_list:
        for a1, lo1, hi1, a2, lo2, hi2, st in rounds:
            grid = capi.Grid(a1, lo1, hi1, a2, lo2, hi2)
            cr, k1, k2 = capi.joint_grid_cells(grid)
            o = oracle.joint_2d(j["region"], j["reads"], cr, k1, k2, read_strand=st)
            has = np.zeros(n, bool); has[cr] = True
            got = {}
            for name, b in (("chained", by_grid), ("mid", by_grid_mid), ("tails", by_grid_tails)):
                assert b.set_grid(grid, st) == len(cr) > 0
                b.run(); b.sync()
                got[name] = b.fetch()
            by_list.set_cells(cr, k1, k2, st)
            by_list.run(); by_list.sync()
            got["list"] = by_list.fetch()
            for name, g in got.items():
                for key in o:
                    sel = has if len(o[key]) == n else slice(None)
                    assert np.array_equal(np.asarray(g[key])[sel], np.asarray(o[key])[sel]), (name, a1, key)
        # a grid that gives no read a cell is an empty round, not an error
        assert by_grid.set_grid(capi.Grid((0, 1, 4), t1 + 50, t1 + 60, (0, 1, 4), t2, t2 + 1), strands) == 0


def test_joint_finer_grid_from_kept_column_states(capi, oracle):
    """A coarse routed grid's sweeps leave a column state at every repeat count of a read's range; a finer grid inside
    (the reference's round 3 after round 2, nanoRepeat_joint.py:275-349) runs no sweep at all -- fewer executed cells,
    the same results as a batch that sweeps again (NRA_F_JOINT_NO_KEEP) and as the oracle.  A grid that leaves the kept
    ranges, other strands, or nra_batch2d_invalidate make the library sweep again."""
    j = synth.make_joint(17, alleles=((11, 6), (21, 4)), read_len=640, read_sd=60, anchor=330, seed=75)
    n = len(j["reads"])
    t1, t2 = j["truth"][:, 0].astype(np.float64), j["truth"][:, 1].astype(np.float64)
    strands = j["strand"].astype(np.int8)
    flipped = strands.copy(); flipped[3] = -flipped[3]
    lo1, hi1, lo2, hi2 = t1 - 9, t1 + 8, np.maximum(t2 - 5, 0), t2 + 5
    coarse = capi.Grid((1, 4, 9), lo1, hi1, (0, 3, 5), lo2, hi2)
    size1, size2 = t1 + np.arange(n) % 3 - 1 + 0.5 * (np.arange(n) % 2), t2 + np.arange(n) % 2      # "round-2 sizes", some of them means of ties
    fine = capi.Grid((0, 1, 40), np.maximum(size1 - 4, lo1), np.minimum(size1 + 4, hi1),
                     (0, 1, 16), np.maximum(size2 - 3, lo2), np.minimum(size2 + 3, hi2))
    wider = capi.Grid((0, 1, 40), lo1 - 3, hi1 + 3, (0, 1, 16), lo2, hi2 + 2)      # beyond what was kept
    steps = [("coarse", coarse, strands, "sweeps"), ("fine", fine, strands, "kept"), ("fine again", fine, strands, "kept"),
             ("fine, one read on the other strand", fine, flipped, "sweeps"), ("fine once more", fine, flipped, "kept"),
             ("wider", wider, flipped, "sweeps"), ("coarse", coarse, strands, "sweeps"), ("invalidate", None, None, None),
             ("fine after invalidate", fine, strands, "sweeps"), ("fine, strands probed", fine, None, "sweeps")]
    with capi.Batch.create_2d_reads(j["region"], j["reads"]) as keeps, \
            capi.Batch.create_2d_reads(j["region"], j["reads"], flags=capi.F_JOINT_NO_KEEP) as sweeps:
        for what, grid, st, expect in steps:
            if grid is None:
                keeps.invalidate()
                continue
            cr, k1, k2 = capi.joint_grid_cells(grid)
            o = oracle.joint_2d(j["region"], j["reads"], cr, k1, k2, read_strand=st)
            has = np.zeros(n, bool); has[cr] = True
            cells = {}
            for name, b in (("keeps", keeps), ("sweeps", sweeps)):
                assert b.set_grid(grid, st) == len(cr) > 0
                b.run(); b.sync()
                g = b.fetch()
                cells[name] = b.stats()["executed_cells"]
                for key in o:
                    sel = has if len(o[key]) == n else slice(None)
                    assert np.array_equal(np.asarray(g[key])[sel], np.asarray(o[key])[sel]), (what, name, key)
            if expect == "kept":
                assert cells["keeps"] < 0.5 * cells["sweeps"], (what, cells)
            else:
                assert cells["keeps"] >= cells["sweeps"], (what, cells)      # (a coarse grid keeps a little more than it needs)


def test_joint_refinement_routed_on_the_device(capi, oracle):
    """nra_batch2d_refine: the reference's round 3 (nanoRepeat_joint.py:275-349) enqueued behind round 2's run and routed
    on the device from round 2's per-read results.  Equal, per read, to the two-call path -- fetch round 2, route the
    finer grid on the host (max(size - s, lo) <= k < min(size + s, hi)), set_grid, run -- and to the oracle on that
    finer grid; the statistics count both grids' cells; a batch without kept column states says NRA_E_STATE."""
    j = synth.make_joint(40, alleles=((11, 6), (21, 4)), read_len=640, read_sd=60, anchor=330, seed=91)
    n = len(j["reads"])
    t1, t2 = j["truth"][:, 0].astype(np.float64), j["truth"][:, 1].astype(np.float64)
    strands = j["strand"].astype(np.int8)
    lo1, hi1, lo2, hi2 = t1 - 9, t1 + 8, np.maximum(t2 - 5, 0), t2 + 5
    lo1[5], hi1[5] = 3.0, 3.0                                  # a read without cells: no round-2 size, no refinement
    junk = list(j["reads"]); junk[7] = "ACGT" * 30             # a read no cell of which reaches min_dp_score
    for s1, s2 in ((4, 3), (2, 2), (5, 2)):
        coarse = capi.Grid((1, s1, 40 // s1), lo1, hi1, (0, s2, 16 // s2 + 1), lo2, hi2)
        with capi.Batch.create_2d_reads(j["region"], junk) as two, capi.Batch.create_2d_reads(j["region"], junk) as one:
            n2 = two.set_grid(coarse, strands)
            two.run(); two.sync()
            r2 = two.fetch(per_candidate=False)
            ok = (r2["status"] == 0) & (r2["n_ties"] > 0)
            assert ok.sum() >= n - 3 and not ok[5] and not ok[7]
            nt = np.maximum(r2["n_ties"], 1).astype(np.float64)
            z1, z2 = r2["sum_k1"] / nt, r2["sum_k2"] / nt
            f = [np.where(ok, v, 0.0) for v in (np.maximum(z1 - s1, lo1), np.minimum(z1 + s1, hi1), np.maximum(z2 - s2, lo2), np.minimum(z2 + s2, hi2))]
            fine = capi.Grid((0, 1, 60), f[0], f[1], (0, 1, 30), f[2], f[3])
            n3 = two.set_grid(fine, strands)
            two.run(); two.sync()
            want = two.fetch(per_candidate=False)
            cr, k1, k2 = capi.joint_grid_cells(fine)
            o = oracle.joint_2d(j["region"], junk, cr, k1, k2, read_strand=strands)
            has = np.zeros(n, bool); has[cr] = True

            assert one.set_grid(coarse, strands) == n2
            one.run()
            assert one.refine(s1, s2, lo1, hi1, lo2, hi2)
            one.sync()
            st = one.stats()
            got = one.fetch()
            assert st["n_alignments"] == n2 + n3 == n2 + len(cr), (s1, s2)
            for key in ("best_wscore", "sum_k1", "sum_k2", "n_ties", "status"):
                assert np.array_equal(got[key], want[key]), (s1, s2, key)
                assert np.array_equal(np.asarray(got[key])[has], np.asarray(o[key])[has]), (s1, s2, key)
            assert (got["status"][~has] == 2).all()
            # the refinement's cells: (2 s1)(2 s2) entries a read, its n1 x n2 cells first (k1-major, like the finer grid's list)
            cap = 4 * s1 * s2
            cs = got["cell_score"].reshape(n, cap); cw = got["cell_wscore"].reshape(n, cap)
            for r in np.nonzero(has)[0]:
                mine = np.nonzero(cr == r)[0]
                assert np.array_equal(cs[r, :len(mine)], o["cell_score"][mine]) and np.array_equal(cw[r, :len(mine)], o["cell_wscore"][mine])
                assert (cs[r, len(mine):] == -1).all()
            # a second refinement of the same run, or one after the run was waited for: not in this state
            assert not one.refine(s1, s2, lo1, hi1, lo2, hi2)
            # the grid itself runs again on the same batch: its own cells and results (the refinement kept arrays of its own)
            one.run(); one.sync()
            rerun = one.fetch()
            assert len(rerun["cell_score"]) == n2
            for key in ("best_wscore", "sum_k1", "sum_k2", "n_ties", "status"):
                assert np.array_equal(rerun[key], r2[key]), (s1, s2, "grid again", key)
            # a refinement behind a grid that itself ran from kept column states (every second value of the coarse grid: no
            # sweep, the template pool still has to reach the last kept count)
            sparse = capi.Grid((1, 2 * s1, 40 // (2 * s1)), lo1, hi1, (0, s2, 16 // s2 + 1), lo2, hi2)
            crs, k1s, k2s = capi.joint_grid_cells(sparse)
            osp = oracle.joint_2d(j["region"], junk, crs, k1s, k2s, read_strand=strands)
            oks = (osp["status"] == 0) & (osp["n_ties"] > 0)
            nts = np.maximum(osp["n_ties"], 1).astype(np.float64)
            y1, y2 = osp["sum_k1"] / nts, osp["sum_k2"] / nts
            fs_ = [np.where(oks, v, 0.0) for v in (np.maximum(y1 - s1, lo1), np.minimum(y1 + s1, hi1), np.maximum(y2 - s2, lo2), np.minimum(y2 + s2, hi2))]
            crf, k1f, k2f = capi.joint_grid_cells(capi.Grid((0, 1, 60), fs_[0], fs_[1], (0, 1, 30), fs_[2], fs_[3]))
            of = oracle.joint_2d(j["region"], junk, crf, k1f, k2f, read_strand=strands)
            hasf = np.zeros(n, bool); hasf[crf] = True
            one.invalidate()                                  # (the coarse grid sweeps again and keeps)
            assert one.set_grid(coarse, strands) == n2
            one.run(); one.sync()
            swept_cells = one.stats()["executed_cells"]
            assert one.set_grid(sparse, strands) == len(crs)
            one.run()
            assert one.refine(s1, s2, lo1, hi1, lo2, hi2)
            one.sync()
            assert one.stats()["executed_cells"] < 0.5 * swept_cells          # no sweep ran
            gf = one.fetch(per_candidate=False)
            for key in ("best_wscore", "sum_k1", "sum_k2", "n_ties", "status"):
                assert np.array_equal(np.asarray(gf[key])[hasf], np.asarray(of[key])[hasf]), (s1, s2, "after reuse", key)
            # the batch goes on: the next grid sweeps or reuses as before
            assert one.set_grid(fine, strands) == n3
            one.run(); one.sync()
            again = one.fetch(per_candidate=False)
            for key in again:
                assert np.array_equal(again[key], want[key]), key
    with capi.Batch.create_2d_reads(j["region"], junk, flags=capi.F_JOINT_NO_KEEP) as b:
        assert b.set_grid(coarse, strands) > 0
        b.run()
        assert not b.refine(s1, s2, lo1, hi1, lo2, hi2)         # nothing kept: the caller takes the two-call path
        b.sync()
    with capi.Batch.create_2d_reads(j["region"], junk) as b:
        assert b.set_grid(coarse, None) > 0                     # strands probed: nothing kept either
        b.run()
        assert not b.refine(s1, s2, lo1, hi1, lo2, hi2)
        b.sync()


def test_kept_column_states_budget_and_the_give_up_word(capi, oracle, monkeypatch):
    """Two guards that should never decide a result.  NRA_JOINT_KEEP_BUDGET_GB = 0: a routed grid keeps no column states,
    a refinement is refused (NRA_E_STATE) and the finer grid sweeps again -- same results.  NRA_TEST_MT_GIVEUP: a run of
    concurrent row blocks that starts with its give-up word set (as if a wave had timed out waiting for the block above
    it) ends at once and the fetch reports NRA_E_DEVICE instead of results."""
    j = synth.make_joint(12, alleles=((11, 6), (21, 4)), read_len=640, read_sd=60, anchor=330, seed=5)
    t1, t2 = j["truth"][:, 0].astype(np.float64), j["truth"][:, 1].astype(np.float64)
    strands = j["strand"].astype(np.int8)
    lo1, hi1, lo2, hi2 = t1 - 9, t1 + 8, np.maximum(t2 - 5, 0), t2 + 5
    coarse = capi.Grid((1, 4, 9), lo1, hi1, (0, 3, 5), lo2, hi2)
    fine = capi.Grid((0, 1, 40), t1 - 2, t1 + 2, (0, 1, 16), np.maximum(t2 - 1, 0), t2 + 2)
    cr, k1, k2 = capi.joint_grid_cells(fine)
    o = oracle.joint_2d(j["region"], j["reads"], cr, k1, k2, read_strand=strands)
    cells = {}
    for budget in (None, "0"):
        if budget is not None:
            monkeypatch.setenv("NRA_JOINT_KEEP_BUDGET_GB", budget)
        with capi.Batch.create_2d_reads(j["region"], j["reads"]) as b:
            assert b.set_grid(coarse, strands) > 0
            b.run()
            assert b.refine(4, 3, lo1, hi1, lo2, hi2) == (budget is None)
            b.sync()
            assert b.set_grid(fine, strands) == len(cr)
            b.run(); b.sync()
            g = b.fetch()
            cells[budget] = b.stats()["executed_cells"]
            for key in o:
                assert np.array_equal(np.asarray(g[key]), np.asarray(o[key])), (budget, key)
    monkeypatch.delenv("NRA_JOINT_KEEP_BUDGET_GB")
    assert cells[None] < 0.5 * cells["0"]
    # 1D, reads beyond one register block: row blocks as concurrent waves
    rng = np.random.default_rng(8)
    left, right = synth.rand_seq(rng, 300), synth.rand_seq(rng, 300)
    reads = [synth.apply_errors(rng, left[-100:] + "TATTG" * k + right[:100], "hifi") for k in (700, 720, 760, 800)]
    kmin, kmax = np.array([690, 710, 750, 790], np.int32), np.array([710, 730, 770, 810], np.int32)
    with capi.Batch.create_1d([(left, "TATTG", right)], reads, kmin, kmax) as b:
        b.run(); b.sync()
        good = b.fetch(per_candidate=False)
        assert (good["status"] == 0).all()
        monkeypatch.setenv("NRA_TEST_MT_GIVEUP", "1")
        b.run()
        with pytest.raises(capi.NraError) as e:
            b.sync()
        assert e.value.code == -2 and "timed out" in str(e.value)
        monkeypatch.delenv("NRA_TEST_MT_GIVEUP")
        b.run(); b.sync()                                     # the batch goes on
        again = b.fetch(per_candidate=False)
        for key in good:
            assert np.array_equal(good[key], again[key]), key


def test_joint_flank_sweeps_ahead_of_the_cell_list(capi, oracle):
    """nra_batch2d_sweep_flanks: the strand-only packed sweeps of L and rev(R) enqueued before any cell list exists.  The
    lists that follow -- routed grids, explicit cells, other strands for some reads, strands left to the probe -- give the
    oracle's results, and the same executed cells as a batch that was never warmed (the work moved, it did not grow)."""
    j = synth.make_joint(23, alleles=((11, 6), (21, 4)), read_len=640, read_sd=60, anchor=330, seed=12)
    n = len(j["reads"])
    t1, t2 = j["truth"][:, 0].astype(np.float64), j["truth"][:, 1].astype(np.float64)
    strands = j["strand"].astype(np.int8)
    partly = strands.copy(); partly[[2, 3, 11]] = 0           # strands round 1 did not find: their pairs are not swept ahead
    flipped = strands.copy(); flipped[4] = -flipped[4]
    grid = capi.Grid((1, 4, 9), t1 - 9, t1 + 8, (0, 3, 5), np.maximum(t2 - 5, 0), t2 + 5)
    cr, k1, k2 = capi.joint_grid_cells(grid)
    for ahead, given in ((strands, strands), (partly, strands), (strands, flipped), (partly, None), (strands, None)):
        o = oracle.joint_2d(j["region"], j["reads"], cr, k1, k2, read_strand=given)
        cells = {}
        for warmed in (True, False):
            with capi.Batch.create_2d_reads(j["region"], j["reads"]) as b:
                if warmed:
                    b.sweep_flanks(ahead)
                    b.sweep_flanks(ahead)                         # (a second call finds nothing left to sweep)
                assert b.set_grid(grid, given) == len(cr)
                b.run(); b.sync()
                g = b.fetch()
                cells[warmed] = b.stats()["executed_cells"]
                for key in o:
                    assert np.array_equal(np.asarray(g[key]), np.asarray(o[key])), (warmed, key)
                # ... and an explicit cell list on the same batch afterwards
                b.set_cells(cr[::2], k1[::2], k2[::2], given)
                b.run(); b.sync()
                g2 = b.fetch()
                o2 = oracle.joint_2d(j["region"], j["reads"], cr[::2], k1[::2], k2[::2], read_strand=given)
                for key in o2:
                    assert np.array_equal(np.asarray(g2[key]), np.asarray(o2[key])), (warmed, "cells", key)
        if given is not None and (ahead == given).all():
            assert cells[True] == cells[False], cells
        else:
            assert cells[True] >= cells[False], cells             # pairs swept ahead for a strand the list then does not use


def test_joint_beyond_the_keep_budget_sweeps_again(capi, monkeypatch):
    """30 000 amplicon reads keep 17 GiB of column states.  With the budget set below that (NRA_JOINT_KEEP_BUDGET_GB = 16) the
    batch keeps nothing and round 3 sweeps again -- cell for cell what a NRA_F_JOINT_NO_KEEP batch computes and executes;
    with the built-in budget (a third of the device) they are kept, as for 2000 reads of the same kind: the same results
    from fewer cells in round 3."""
    j = synth.make_joint(30000, seed=3)
    t1, t2 = j["truth"][:, 0].astype(np.float64), j["truth"][:, 1].astype(np.float64)
    strands = j["strand"].astype(np.int8)
    coarse = capi.Grid((0, 4, 25), t1 - 20, t1 + 13, (0, 3, 8), np.zeros(len(t2)), t2 + 8)       # 33 + 15..18 counts kept per read: 0.8 MB
    fine = capi.Grid((0, 1, 90), t1 - 2, t1 + 3, (0, 1, 24), np.maximum(t2 - 2, 0), t2 + 2)
    for n, budget, keeps in ((30000, "16", False), (30000, None, True), (2000, None, True)):
        sub = lambda g: capi.Grid(g.axes[0], g.bounds[0][:n], g.bounds[1][:n], g.axes[1], g.bounds[2][:n], g.bounds[3][:n])
        if budget is None:
            monkeypatch.delenv("NRA_JOINT_KEEP_BUDGET_GB", raising=False)
        else:
            monkeypatch.setenv("NRA_JOINT_KEEP_BUDGET_GB", budget)
        got = {}
        for name, flags in (("default", 0), ("no keep", capi.F_JOINT_NO_KEEP)):
            with capi.Batch.create_2d_reads(j["region"], j["reads"][:n], flags=flags) as b:
                for grid in (sub(coarse), sub(fine)):
                    assert b.set_grid(grid, strands[:n]) > 0
                    b.run(); b.sync()
                    got.setdefault(name, []).append((b.fetch(), b.stats()["executed_cells"]))
        for (a, cells_a), (c, cells_c) in zip(got["default"], got["no keep"]):
            for key in a:
                assert np.array_equal(a[key], c[key]), (n, key)
        (_, r2_default), (_, r3_default) = got["default"]
        (_, r2_nokeep), (_, r3_nokeep) = got["no keep"]
        if keeps:
            assert r2_default >= r2_nokeep and r3_default < 0.5 * r3_nokeep, (n, got["default"][1][1], r3_nokeep)
        else:
            assert (r2_default, r3_default) == (r2_nokeep, r3_nokeep), n
    monkeypatch.delenv("NRA_JOINT_KEEP_BUDGET_GB", raising=False)


def test_joint_packed_flank_sweeps(capi, oracle):
    """The columns of L and rev(R) outside the scoring window are swept in packed int16 cells, two reads per
    wave, and the int32 sweeps resume from the state they leave: flanks just below / at / above the 64-column
    threshold, N in flanks and reads, an odd number of reads, several row buckets, a scoring scheme the packed
    cells cannot hold (falls back to int32 cells)."""
    rng = np.random.default_rng(97)
    u1, u2, mid = "CAG", "CCG", "CAACAGCCGCCAC"
    for fl, fr, sc_over in ((73, 300, None), (74, 74, None), (200, 90, None), (150, 150, dict(mismatch=9)),
                            (300, 120, dict(match=3, gap_open2=30))):
        L, R = synth.rand_seq(rng, fl), synth.rand_seq(rng, fr)
        if fl >= 150:
            L = L[:40] + "N" + L[41:120] + "NN" + L[122:]
        reads, cr, k1, k2 = [], [], [], []
        for r in range(7):
            a, b = int(rng.integers(3, 20)), int(rng.integers(2, 9))
            cut_l, cut_r = int(rng.integers(0, fl // 2)), int(rng.integers(0, fr // 2))
            s = synth.apply_errors(rng, L[cut_l:] + u1 * a + mid + u2 * b + R[:fr - cut_r], "ont")
            if r == 2:
                s = s[:30] + "N" + s[31:]
            if r == 5:
                s = synth.rand_seq(rng, 900) + s              # another rows-per-lane bucket
            if rng.random() < 0.5:
                s = synth.revcomp(s)
            reads.append(s)
            for x in range(max(0, a - 3), a + 3):
                for y in range(max(0, b - 2), b + 3, 2):
                    cr.append(r); k1.append(x); k2.append(y)
        o = oracle.joint_2d((L, u1, mid, u2, R), reads, cr, k1, k2, sc=oracle.default_scoring(**(sc_over or {})))
        for flags in (0, capi.F_NO_JOINT_PACK):
            g = capi.joint_2d((L, u1, mid, u2, R), reads, cr, k1, k2, sc=capi.default_scoring(**(sc_over or {})), flags=flags)
            for k in KEYS_2D:
                assert np.array_equal(g[k], o[k]), (fl, fr, flags, k, np.nonzero(g[k] != o[k])[0][:8])


def test_more_chained_tasks_than_scratch_strips(capi, oracle):
    """A chained launch has one wave per scratch strip and walks its tasks with a grid stride: 9000 small reads
    in forced chained blocks (NRA_F_TEST_CHAIN) are 4500 packed LDS-ring tasks (> 4096 strips) and, with the
    DPP sweeps, 9000 int32 tasks (> 512 strips).  Same results as the unchained sweeps; a sample equals the oracle."""
    d = synth.make_1d(9000, "CAG", (6, 11), "ont_q20", kwin=(2, 16), anchor=40, flank=25, seed=88)
    base = capi.round3_1d(d["regions"], d["reads"], d["kmin"], d["kmax"])
    for flags in (capi.F_TEST_CHAIN, capi.F_TEST_CHAIN | capi.F_SERIAL_CHAIN, capi.F_TEST_CHAIN | capi.F_DPP_SWEEP):
        g = capi.round3_1d(d["regions"], d["reads"], d["kmin"], d["kmax"], flags=flags)
        for k in ("best_score", "sum_k", "n_ties", "status", "cand_score"):
            assert np.array_equal(g[k], base[k]), (flags, k)
    pick = np.arange(0, 9000, 450)
    o = oracle.round3_1d(d["regions"], [d["reads"][i] for i in pick], d["kmin"][pick], d["kmax"][pick])
    for k in ("best_score", "sum_k", "n_ties", "status"):
        assert np.array_equal(base[k][pick], o[k]), k


def test_large_copies_through_the_pinned_stage(capi):
    """The library copies 896 KB and more through a pinned stage of its own, in pieces of 2 MB (copy_h2d / copy_d2h,
    nra_host.cpp): one batch of 5400 config-2 reads -- 4.2 MB per candidate array down, three pieces with a ragged last one --
    against the same reads in batches of 300, whose copies all stay below the threshold and go straight to hipMemcpy; and
    sizes just below / at the threshold."""
    d = synth.config2(n_reads=5400)
    with capi.Batch.create_1d(d["regions"], d["reads"], d["kmin"], d["kmax"], flags=capi.F_TIE_EXTENTS) as b:
        b.run(); b.sync()
        big = b.fetch()
    assert big["cand_score"].nbytes > (4 << 20)
    keys = ("best_score", "sum_k", "n_ties", "status", "cand_score", "cand_tstart", "cand_tend")
    parts = []
    for a in range(0, 5400, 300):
        sl = slice(a, a + 300)
        with capi.Batch.create_1d(d["regions"], d["reads"][sl], d["kmin"][sl], d["kmax"][sl], flags=capi.F_TIE_EXTENTS) as b:
            b.run(); b.sync()
            parts.append(b.fetch())
        assert parts[-1]["cand_score"].nbytes < (256 << 10)
    for k in keys:
        assert np.array_equal(big[k], np.concatenate([p[k] for p in parts])), k
    for n in (1170, 1171):                                  # 196 candidates x 4 B a read: 917 280 / 918 064 B either side of 896 KB
        sl = slice(0, n)
        with capi.Batch.create_1d(d["regions"], d["reads"][sl], d["kmin"][sl], d["kmax"][sl]) as b:
            b.run(); b.sync()
            got = b.fetch()
        assert np.array_equal(got["cand_score"], big["cand_score"][:len(got["cand_score"])]), n


def test_large_call_streams_region_blocks(capi, oracle):
    """nra_round3_1d cuts a call of >= 131072 reads grouped by region into region blocks and packs / uploads block
    i + 1 while block i's kernels run: same results as the one resident batch, per read and per candidate, also
    when a region straddles the nominal block boundary; a sample equals the oracle."""
    rng = np.random.default_rng(5)
    regions, reads, rr, kmin, kmax = [], [], [], [], []
    sizes = [20000, 300, 65000, 7, 48000, 1, 10000]            # 143 308 reads in 7 regions of very different size
    for g, n in enumerate(sizes):
        unit = ("CAG", "TATTG", "AT", "GGC", "CAG", "AC", "TTTA")[g]
        d = synth.make_1d(400, unit, (5, 9), "ont_q20", kwin=(2, 12), anchor=30, flank=20, rng=rng)
        regions += d["regions"]
        pick = rng.integers(0, 400, size=n)
        reads += [d["reads"][i] for i in pick]; rr += [g] * n
        kmin += [2] * n; kmax += [12] * n
    with capi.Batch.create_1d(regions, reads, kmin, kmax, read_region=rr) as b:
        b.run(); b.sync()
        want = b.fetch()
    got = capi.round3_1d(regions, reads, kmin, kmax, read_region=rr)
    for k in ("best_score", "sum_k", "n_ties", "status", "cand_score"):
        assert np.array_equal(got[k], want[k]), k
    pick = np.arange(0, len(reads), 4999)
    o = oracle.round3_1d(regions, [reads[i] for i in pick], [kmin[i] for i in pick], [kmax[i] for i in pick],
                         read_region=[rr[i] for i in pick])
    for k in ("best_score", "sum_k", "n_ties", "status"):
        assert np.array_equal(got[k][pick], o[k]), k


def test_row_blocks_as_concurrent_waves(capi, oracle):
    """k_sweep_ringmt: the row blocks of a long read are waves of their own that hand the columns down through granule
    strips.  Packed (two reads per wave, 3.2 - 6.5 kb: 4 - 7 blocks of 960 rows) and int32 cells (> 6750 bases) in one
    batch, reads of different block counts paired, a lone read, tight and wide windows (first boundary early and
    late in the sweep), the same batch run twice (fresh epochs), and against the one-wave-per-read chain."""
    rng = np.random.default_rng(31)
    L, R = synth.rand_seq(rng, 600), synth.rand_seq(rng, 500)
    reads, kmin, kmax = [], [], []
    for k_true, lo, hi in ((600, 590, 612), (640, 500, 660), (900, 880, 915), (1250, 1247, 1253), (700, 699, 701),
                           (1500, 1490, 1512), (610, 0, 30)):
        reads.append(synth.apply_errors(rng, L[-100:] + "TATTG" * k_true + R[:100], "ont_q20"))
        kmin.append(lo); kmax.append(hi)
    assert min(len(r) for r in reads) > 3072 and max(len(r) for r in reads) > 6750
    o = oracle.round3_1d([(L, "TATTG", R)], reads, kmin, kmax)
    with capi.Batch.create_1d([(L, "TATTG", R)], reads, kmin, kmax) as b:
        for _ in range(2):
            b.run(); b.sync()
            g = b.fetch()
            for k in ("best_score", "sum_k", "n_ties", "status", "cand_score"):
                assert np.array_equal(g[k], o[k]), (k, g[k][:8], o[k][:8])
    s = capi.round3_1d([(L, "TATTG", R)], reads, kmin, kmax, flags=capi.F_SERIAL_CHAIN)
    for k in ("best_score", "sum_k", "n_ties", "status", "cand_score"):
        assert np.array_equal(s[k], o[k]), k


def test_row_block_heights_and_the_choice_for_2_to_3_kb_reads(capi, oracle):
    """Reads of 2049 - 3072 bases run as row blocks when that is the cheaper form for the batch (a few reads: always),
    in blocks of 64 x 12 .. 15 rows -- the height that pads the bucket least: one batch per height here, each against the
    oracle, against one register block per read (NRA_CHAIN_FROM=3072), and with the executed cells dist.py predicts."""
    import os
    from nanorepeat_amd import dist as D
    rng = np.random.default_rng(58)
    L, R = synth.rand_seq(rng, 300), synth.rand_seq(rng, 260)
    region = [(L, "TATTG", R)]
    for k_true, rows_per_lane in ((410, 12), (450, 13), (490, 14), (530, 15)):
        reads, kmin, kmax = [], [], []
        for i in range(6):                  # (pairs of reads share a wave: an even number, for the cost model's sake)
            reads.append(synth.apply_errors(rng, L[-90:] + "TATTG" * (k_true + i % 5) + R[:90], "hifi"))
            kmin.append(k_true - 3); kmax.append(k_true + 6)
        q = [len(r) for r in reads]
        assert 64 * rows_per_lane * 2 < min(q) and max(q) <= 64 * rows_per_lane * 3, q
        o = oracle.round3_1d(region, reads, kmin, kmax)
        executed = {}
        for form, env in (("blocks", None), ("one block", "3072")):
            if env is None:
                os.environ.pop("NRA_CHAIN_FROM", None)
            else:
                os.environ["NRA_CHAIN_FROM"] = env
            try:
                with capi.Batch.create_1d(region, reads, kmin, kmax) as b:
                    b.run(); b.sync()
                    g = b.fetch(); executed[form] = b.stats()["executed_cells"]
            finally:
                os.environ.pop("NRA_CHAIN_FROM", None)
            for k in ("best_score", "sum_k", "n_ties", "status", "cand_score"):
                assert np.array_equal(g[k], o[k]), (k_true, form, k)
        # 3 blocks of the expected height against 40 / 48 rows per lane in one block
        ratio = executed["blocks"] * (48 if max(q) > 2560 else 40) / (executed["one block"] * 3 * rows_per_lane)
        assert abs(ratio - 1) < 0.01, (k_true, executed)
        model = int(D.executed_cells(region, q, kmax, [0] * 6).sum())
        assert abs(model - executed["blocks"]) <= 0.02 * executed["blocks"], (k_true, model, executed)


def test_one_read_against_a_megabase_template(capi, oracle):
    """A template far beyond the 16-bit extents of the int32 cells (1.2 M columns: k = 240 000 units of 5 bases):
    the read runs as a chained int32 sweep and its ties through the int64 extents kernel.  The scratch strips of
    the chained kernels are sized by the tasks there are (one here), not by the launch's largest grid -- at
    512 / 4096 strips this template would ask for tens of GB.  Beside it a short-template read in the same call."""
    rng = np.random.default_rng(4)
    L, R = synth.rand_seq(rng, 300), synth.rand_seq(rng, 300)
    reads = [synth.apply_errors(rng, L[-100:] + "TATTG" * 80 + R[:100], "ont_q20"),
             synth.apply_errors(rng, L[-100:] + "TATTG" * 30 + R[:100], "ont_q20")]
    kmin, kmax = [239999, 25], [240000, 35]
    o = oracle.round3_1d([(L, "TATTG", R)], reads, kmin, kmax)
    for flags in (0, capi.F_TIE_EXTENTS):
        g = capi.round3_1d([(L, "TATTG", R)], reads, kmin, kmax, flags=flags)
        for k in ("best_score", "sum_k", "n_ties", "status", "cand_score"):
            assert np.array_equal(g[k], o[k]), (flags, k, g[k][:4], o[k][:4])
        if flags:
            ties = o["cand_tstart"] >= 0
            assert np.array_equal(g["cand_tstart"][ties], o["cand_tstart"][ties])
            assert np.array_equal(g["cand_tend"][ties], o["cand_tend"][ties])
    assert g["status"][1] == 0 and g["sum_k"][1] == 30 * g["n_ties"][1]
    capi.release_cached_memory()


def test_long_joint_reads_uncut_equal_oracle(capi, oracle):
    """Joint reads of 5 kb (beyond one register block) are scored uncut, cell by cell in chained row
    blocks with int64 cells; strands probed (0) or given; short reads of the same batch take the sweeps."""
    j = synth.make_joint(6, alleles=((30, 8), (60, 5)), read_len=5200, read_sd=150, anchor=3000, seed=41)
    k = synth.make_joint(4, alleles=((30, 8), (60, 5)), read_len=900, read_sd=40, anchor=3000, seed=41)
    assert j["region"] == k["region"] and min(len(r) for r in j["reads"]) > 3072
    reads = j["reads"] + k["reads"]
    truth = np.concatenate([j["truth"], k["truth"]])
    cr, k1, k2 = [], [], []
    for r in range(len(reads)):
        for a in range(int(truth[r][0]) - 4, int(truth[r][0]) + 5, 2):
            for b in range(int(truth[r][1]) - 2, int(truth[r][1]) + 3, 2):
                cr.append(r); k1.append(a); k2.append(b)
    o = oracle.joint_2d(j["region"], reads, cr, k1, k2)
    g = capi.joint_2d(j["region"], reads, cr, k1, k2)
    for key in o:
        assert np.array_equal(g[key], o[key]), (key, g[key][:10], o[key][:10])
    g2 = capi.joint_2d(j["region"], reads, cr, k1, k2, read_strand=o["read_strand"])
    for key in o:
        assert np.array_equal(g2[key], o[key]), key
    assert set(o["read_strand"][:6].tolist()) == {1, -1} and (o["status"] == 0).all()


def test_config2_order_and_batching_invariance(capi, config2_run):
    d, g, _ = config2_run
    n = len(d["reads"])
    perm = np.random.default_rng(2).permutation(n)[:3000]
    sub = capi.round3_1d(d["regions"], [d["reads"][i] for i in perm], d["kmin"][perm], d["kmax"][perm],
                         per_candidate=False)
    for k in ("best_score", "sum_k", "n_ties", "status"):
        assert np.array_equal(sub[k], g[k][perm]), k


# ------------------------------------------------------------------ junction decomposition vs brute force
def _modes_agree(capi, oracle, d, sc_over=None):
    """Every execution mode of the 1D path gives the oracle's per-read and per-candidate results:
    0 = junction decomposition, flank verdict from three scores (extents DP only when ambiguous);
    TIE_EXTENTS = decomposition + explicit extents for every tie; BRUTE_FORCE = K independent
    alignments; ALL_EXTENTS = explicit extents for every candidate."""
    sc_g = capi.default_scoring(**(sc_over or {}))
    o = oracle.round3_1d(d["regions"], d["reads"], d["kmin"], d["kmax"], read_region=d.get("read_region"),
                         sc=oracle.default_scoring(**(sc_over or {})))
    res = {}
    for flags in (0, capi.F_TIE_EXTENTS, capi.F_BRUTE_FORCE, capi.F_ALL_EXTENTS, capi.F_DPP_SWEEP, capi.F_NO_HALF_WAVE):
        with capi.Batch.create_1d(d["regions"], d["reads"], d["kmin"], d["kmax"],
                                  read_region=d.get("read_region"), sc=sc_g, flags=flags) as b:
            b.run(); b.sync()
            g = b.fetch()
            res[flags] = (g, b.stats())
        for k in ("best_score", "sum_k", "n_ties", "status", "cand_score"):
            assert np.array_equal(g[k], o[k]), (flags, k, np.nonzero(g[k] != o[k])[0][:8], g[k][:12], o[k][:12])
    # explicit extents, where a mode computed them, are the oracle's
    g2 = res[capi.F_TIE_EXTENTS][0]
    assert np.array_equal(g2["cand_tstart"], o["cand_tstart"]) and np.array_equal(g2["cand_tend"], o["cand_tend"])
    return res


def test_1d_decomposition_modes_agree(capi, oracle):
    d = synth.make_1d(24, "TATTG", (8, 30), "ont", kwin=(0, 45), anchor=300, seed=101)
    res = _modes_agree(capi, oracle, d)
    st_fast, st_brute = res[0][1], res[capi.F_BRUTE_FORCE][1]
    assert st_fast["executed_cells"] * 5 < st_brute["executed_cells"]
    assert st_fast["algorithmic_cells"] == st_brute["algorithmic_cells"]
    # different windows per read (reference rule), odd read count, several motifs
    # unit lengths 1..8 run the LDS-ring sweeps (skew = m), longer units the DPP sweeps
    for unit, seed in (("CAG", 102), ("AT", 103), ("GGCCCC", 104), ("A", 105), ("ACGGTCA", 107), ("ACGGTCAT", 108),
                       ("ACGGTCATG", 109), ("ACGGTCATGCAT", 110)):
        _modes_agree(capi, oracle, synth.make_1d(15, unit, (4, 27), "ont_q20", kwin=None, anchor=200, flank=70, seed=seed))
    _modes_agree(capi, oracle, synth.make_1d(9, "TATTG", (12, 20), "hifi", kwin=(10, 22), anchor=150, flank=60, seed=106),
                 sc_over=dict(match=1, mismatch=3, gap_open1=5, gap_ext1=2, gap_open2=20, gap_ext2=1, min_dp_score=20))


def test_1d_decomposition_adversarial(capi, oracle):
    """Reads that break the flank tests or sit on the junction: no left / right flank, flank-only
    reads, a read ending exactly at the repeat end, gaps spanning the junction, N bases, junk,
    1-base flanks, one read per region (unpaired halves)."""
    rng = np.random.default_rng(77)
    L, R, u = synth.rand_seq(rng, 180), synth.rand_seq(rng, 160), "TATTG"
    core = lambda k, fl=70, fr=70: L[len(L) - fl:] + u * k + R[:fr]
    reads = [
        core(9), core(9, 70, 0), core(9, 0, 70), core(9, 70, 1), core(9, 1, 70), core(9, 70, 3),
        L[-70:], R[:70], u * 12, core(0), core(1),
        core(9)[:70 + 45 - 7] + core(9)[70 + 45 + 6:],      # deletion across the unit/R junction
        core(9)[:70 + 45] + "ACGTTGCAAC" + core(9)[70 + 45:],  # insertion at the junction
        core(9)[:70 + 45 - 2] + "N" * 4 + core(9)[70 + 45 + 2:],
        core(7) + synth.rand_seq(rng, 60), synth.rand_seq(rng, 60) + core(7),
        synth.rand_seq(rng, 150), "ACG", "",
        core(14), core(15), core(16),
    ]
    n = len(reads)
    d = dict(regions=[(L, u, R)], reads=reads, kmin=np.zeros(n, np.int32), kmax=np.full(n, 20, np.int32))
    d["kmin"][3] = 5; d["kmax"][3] = 4          # skipped read among paired ones
    _modes_agree(capi, oracle, d)
    _modes_agree(capi, oracle, d, sc_over=dict(min_dp_score=0))
    # 1-base flanks and a flank-less region (falls back to brute force inside the library)
    for Lx, Rx in ((L[-1:], R[:1]), (L[-1:], R), ("", R), (L, "")):
        dx = dict(regions=[(Lx, u, Rx)], reads=[u * 7, Lx + u * 5 + Rx, "GG" + u * 6 + "TT"], kmin=[0, 0, 2], kmax=[12, 9, 9])
        _modes_agree(capi, oracle, dx, sc_over=dict(min_dp_score=10))
    # many regions with one or two reads each
    regions, rds, rr, kmin, kmax = [], [], [], [], []
    for g in range(7):
        unit = synth.rand_unit(rng, 2 + g % 5)
        Lg, Rg = synth.rand_seq(rng, 60 + 7 * g), synth.rand_seq(rng, 90 - 5 * g)
        regions.append((Lg, unit, Rg))
        for j in range(1 + g % 2):
            k = int(rng.integers(0, 25))
            rds.append(synth.apply_errors(rng, Lg[-50:] + unit * k + Rg[:50], "ont")); rr.append(g)
            kmin.append(max(0, k - 6)); kmax.append(k + 6 + j)
    _modes_agree(capi, oracle, dict(regions=regions, reads=rds, kmin=kmin, kmax=kmax, read_region=rr),
                 sc_over=dict(min_dp_score=30))


# ------------------------------------------------------------------ the other BASELINE configs (reduced)
def test_config5_wide_sweep_sample(capi, oracle):
    """Config 5 shape: HiFi error model, k in [5,500] (496 candidates), 2.3 kb cores (R = 40)."""
    d = synth.config5(n_reads=6, seed=55)
    g = capi.round3_1d(d["regions"], d["reads"], d["kmin"], d["kmax"])
    o = oracle.round3_1d(d["regions"], d["reads"], d["kmin"], d["kmax"])
    for k in KEYS_1D:
        assert np.array_equal(g[k], o[k]), k
    assert (g["status"] == 0).all()
    assert np.array_equal(g["sum_k"] // g["n_ties"], d["k_true"])
    gb = capi.round3_1d(d["regions"], d["reads"][:2], d["kmin"][:2], d["kmax"][:2], flags=capi.F_BRUTE_FORCE)
    for k in KEYS_1D:
        assert np.array_equal(gb[k][:2 if "cand" not in k else 2 * 496], o[k][:2 if "cand" not in k else 2 * 496]), k


def test_config4_many_regions_sample(capi, oracle):
    """Config 4 shape: many regions, mixed 3-6 bp motifs, the reference's window rule."""
    d = synth.config4(n_regions=12, reads_per_region=25, seed=44)
    g = capi.round3_1d(d["regions"], d["reads"], d["kmin"], d["kmax"], read_region=d["read_region"])
    o = oracle.round3_1d(d["regions"], d["reads"], d["kmin"], d["kmax"], read_region=d["read_region"])
    for k in KEYS_1D:
        assert np.array_equal(g[k], o[k]), k
    ok = g["status"] == 0
    assert ok.mean() > 0.95
    # the same reads shuffled across the batch give the same per-read answers
    perm = np.random.default_rng(4).permutation(len(d["reads"]))
    gp = capi.round3_1d(d["regions"], [d["reads"][i] for i in perm], d["kmin"][perm], d["kmax"][perm],
                        read_region=d["read_region"][perm], per_candidate=False)
    for k in ("best_score", "sum_k", "n_ties", "status"):
        assert np.array_equal(gp[k], g[k][perm]), k


def test_shard_cost_model_follows_the_kernels(capi):
    """dist.executed_cells -- what the sharding balances -- against the cells the batch says its kernels execute
    (nra_stats_t.executed_cells), on a batch mixing half-wave, full-wave and chained reads, short and long units."""
    from nanorepeat_amd import dist as D
    rng = np.random.default_rng(21)
    regions, reads, rr, kmin, kmax = [], [], [], [], []
    for g, (unit, anchor, alleles, n) in enumerate((("TATTG", 1000, (40, 150), 300), ("CAG", 400, (20, 300), 300),
                                                    ("AT", 200, (30, 700), 200), ("ACGTTGCATTAC", 300, (8, 20), 60),
                                                    ("TATTG", 600, (800,), 6))):
        d = synth.make_1d(n, unit, alleles, "ont_q20", kwin=None, anchor=anchor, rng=rng)
        regions += d["regions"]; reads += d["reads"]; rr += [g] * n
        kmin += d["kmin"].tolist(); kmax += d["kmax"].tolist()
    qlen = [len(r) for r in reads]
    assert min(qlen) < 300 and max(qlen) > 3072
    with capi.Batch.create_1d(regions, reads, kmin, kmax, read_region=rr) as b:
        b.run(); b.sync()
        st = b.stats()
    model = int(D.executed_cells(regions, qlen, kmax, rr).sum())
    assert abs(model - st["executed_cells"]) <= 0.02 * st["executed_cells"], (model, st["executed_cells"])


def test_config3_joint_sample(capi, oracle):
    """Config 3 shape: HTT-like CAG+CCG joint grid through the host mirror (both rounds)."""
    j = synth.make_joint(6, seed=33)                  # alleles (17,10) / (55,7), 1.2 kb amplicon reads
    init = J.Round1Estimation()
    fq = {}
    for i, s in enumerate(j["reads"]):
        init.repeat1_count_range_dict[f"r{i}"] = tuple(int(x) for x in j["range1"][i])
        init.repeat2_count_range_dict[f"r{i}"] = tuple(int(x) for x in j["range2"][i])
        fq[f"r{i}"] = f"@r{i}\n{s}\n+\n{'!' * len(s)}\n"
    left, u1, mid, u2, right = j["region"]
    chrom = left + u1 * 19 + mid + u2 * 7 + right
    a = J.Repeat().init_from_string(f"chr4:{len(left)}:{len(left) + 57}:{u1}:200")
    b = J.Repeat().init_from_string(f"chr4:{len(left) + 57 + len(mid)}:{len(left) + 57 + len(mid) + 21}:{u2}:20")
    a.max_size += 10; b.max_size += 10
    import copy
    res = {}
    for name, scorer in (("gpu", None), ("oracle", oracle.joint_2d)):
        fin = J.fine_tune_read_count(init, fq, chrom, copy.deepcopy(a), copy.deepcopy(b), scorer=scorer)
        res[name] = ({k: float(v) for k, v in fin.repeat1_count_dict.items()},
                     {k: float(v) for k, v in fin.repeat2_count_dict.items()}, fin.step_size1, fin.step_size2)
    assert res["gpu"] == res["oracle"]
    # the reads as two / four groups in parallel host threads (what fine_tune_read_count does from 2000 reads on),
    # also on a resident session used twice
    for parts in (2, 4):
        with J.GridSession(J._joint_region(chrom, a, b), fq, parts=parts) as sess:
            assert len(sess.subs) == parts
            for _ in range(2):
                sess.new_run()
                fin = J.fine_tune_read_count(init, fq, chrom, copy.deepcopy(a), copy.deepcopy(b), session=sess)
                got = ({k: float(v) for k, v in fin.repeat1_count_dict.items()},
                       {k: float(v) for k, v in fin.repeat2_count_dict.items()}, fin.step_size1, fin.step_size2)
                assert got == res["gpu"] and list(got[0]) == list(res["gpu"][0])
    k1 = np.array([res["gpu"][0][f"r{i}"] for i in range(6)]); k2 = np.array([res["gpu"][1][f"r{i}"] for i in range(6)])
    assert np.mean(np.abs(k1 - j["truth"][:, 0]) <= 1) >= 0.8 and np.mean(np.abs(k2 - j["truth"][:, 1]) <= 1) >= 0.8


# ------------------------------------------------------------------ long reads: chained row blocks
def test_1d_row_block_chaining_small_blocks(capi, oracle):
    """The chaining mechanism (a read swept as consecutive row blocks in one wave, hand-off through a
    scratch strip) exercised with 128-row blocks on ordinary reads: 2-8 blocks per read."""
    for unit, seed, fl in (("TATTG", 201, 100), ("CAG", 202, 60), ("AT", 203, 30)):
        d = synth.make_1d(14, unit, (6, 31), "ont", kwin=None, anchor=220, flank=fl, seed=seed)
        o = oracle.round3_1d(d["regions"], d["reads"], d["kmin"], d["kmax"])
        for flags in (capi.F_TEST_CHAIN, capi.F_TEST_CHAIN | capi.F_TIE_EXTENTS, capi.F_TEST_CHAIN | capi.F_SERIAL_CHAIN):
            with capi.Batch.create_1d(d["regions"], d["reads"], d["kmin"], d["kmax"], flags=flags) as b:
                b.run(); b.sync(); g = b.fetch()
            keys = KEYS_1D if flags & capi.F_TIE_EXTENTS else ("best_score", "sum_k", "n_ties", "status", "cand_score")
            for k in keys:
                assert np.array_equal(g[k], o[k]), (unit, flags, k)
    # adversarial reads (no flanks, ties on the junction) force the ambiguous-verdict extents DP through the chain too
    rng = np.random.default_rng(9)
    L, R, u = synth.rand_seq(rng, 150), synth.rand_seq(rng, 150), "TATTG"
    reads = [L[-70:] + u * 9 + R[:70], L[-70:] + u * 9, u * 12 + R[:60], L[-70:] + u * 9 + R[:1], u * 30,
             L[-70:] + u * 40 + R[:70]]
    d = dict(regions=[(L, u, R)], reads=reads, kmin=np.zeros(6, np.int32), kmax=np.full(6, 45, np.int32))
    o = oracle.round3_1d(d["regions"], d["reads"], d["kmin"], d["kmax"])
    with capi.Batch.create_1d(d["regions"], d["reads"], d["kmin"], d["kmax"], flags=capi.F_TEST_CHAIN) as b:
        b.run(); b.sync(); g = b.fetch(); st = b.stats()
    for k in ("best_score", "sum_k", "n_ties", "status", "cand_score"):
        assert np.array_equal(g[k], o[k]), k


def test_1d_long_reads_over_one_register_block(capi, oracle):
    """Reads of 3.3-7.9 kb (beyond the 3072 rows a wave holds in registers): 1536-row chained blocks."""
    rng = np.random.default_rng(31)
    L, R, u = synth.rand_seq(rng, 120), synth.rand_seq(rng, 120), "GGCCCC"
    reads, kmin, kmax = [], [], []
    for k in (530, 800, 1290):                      # cores of 3.3, 4.9, 7.9 kb
        reads.append(synth.apply_errors(rng, L[-60:] + u * k + R[:60], "hifi"))
        kmin.append(k - 2); kmax.append(k + 2)
    reads.append(synth.apply_errors(rng, L[-60:] + u * 20 + R[:60], "hifi")); kmin.append(18); kmax.append(23)
    g = capi.round3_1d([(L, u, R)], reads, kmin, kmax)
    o = oracle.round3_1d([(L, u, R)], reads, kmin, kmax)
    for k in KEYS_1D:
        assert np.array_equal(g[k], o[k]), k
    assert (g["status"] == 0).all()
    with pytest.raises(capi.NraError) as e:
        capi.round3_1d([(L, u, R)], reads[:1], kmin[:1], kmax[:1], flags=capi.F_BRUTE_FORCE)
    assert e.value.code == -3


# ------------------------------------------------------------------ 2D junction decomposition
def test_2d_decomposition_matches_oracle_and_brute_force(capi, oracle):
    """The (read, k1)-row sweeps give the oracle's per-cell (score, window score) and per-read
    selection, like the brute-force per-cell kernel, for grid-ordered and arbitrary cell lists."""
    for seed, alleles, anchor in ((61, ((6, 4), (11, 3)), 300), (62, ((17, 10), (30, 7)), 1000), (63, ((3, 9), (8, 2)), 40)):
        j = synth.make_joint(8, alleles=alleles, read_len=500 if anchor < 1000 else 800, read_sd=30, anchor=anchor, seed=seed)
        cr, k1, k2 = _cells(j, step=2)
        o = oracle.joint_2d(j["region"], j["reads"], cr, k1, k2)
        for flags in (0, capi.F_BRUTE_FORCE, capi.F_NO_JOINT_PACK):
            g = capi.joint_2d(j["region"], j["reads"], cr, k1, k2, flags=flags)
            for k in KEYS_2D:
                assert np.array_equal(g[k], o[k]), (seed, flags, k, np.nonzero(g[k] != o[k])[0][:8], g[k][:10], o[k][:10])
        # state buffer reused group by group (one read per group), and all strands given (no probe)
        g2 = capi.joint_2d(j["region"], j["reads"], cr, k1, k2, flags=capi.F_TEST_CHAIN)
        g3 = capi.joint_2d(j["region"], j["reads"], cr, k1, k2, read_strand=o["read_strand"].copy())
        for k in KEYS_2D:
            assert np.array_equal(g2[k], o[k]) and np.array_equal(g3[k], o[k]), (seed, k)
    # cells of a read in arbitrary order (runs of length 1, descending k2, repeated cells)
    j = synth.make_joint(5, alleles=((6, 4), (11, 3)), read_len=450, read_sd=20, anchor=200, seed=64)
    rng = np.random.default_rng(6)
    cr, k1, k2 = [], [], []
    for r in range(5):
        cells = [(int(a), int(b)) for a in range(2, 14, 3) for b in range(0, 8, 2)]
        rng.shuffle(cells)
        cells += cells[:3]
        for a, b in cells:
            cr.append(r); k1.append(a); k2.append(b)
    o = oracle.joint_2d(j["region"], j["reads"], cr, k1, k2)
    g = capi.joint_2d(j["region"], j["reads"], cr, k1, k2)
    for k in KEYS_2D:
        assert np.array_equal(g[k], o[k]), k
    # N bases, very short flanks (|R| = 2 is the smallest the decomposition takes; |R| = 1 falls back)
    left, u1, mid, u2, right = j["region"]
    for Lx, Rx in ((left[-12:], right[:2]), (left[-3:], right[:11]), (left[-1:], right[:1])):
        reads = [Lx + u1 * 7 + mid + u2 * 5 + Rx, synth.revcomp(Lx + u1 * 9 + mid + u2 * 3 + Rx),
                 "ACGTNNACGT" + u1 * 5 + "N" + mid + u2 * 4 + Rx]
        cr2 = [r for r in range(3) for _ in range(20)]
        a2 = [a for _ in range(3) for a in range(3, 13, 2) for _ in range(4)]
        b2 = [b for _ in range(3) for _ in range(5) for b in range(1, 9, 2)]
        o = oracle.joint_2d((Lx, u1, mid, u2, Rx), reads, cr2, a2, b2, sc=oracle.default_scoring(min_dp_score=20))
        g = capi.joint_2d((Lx, u1, mid, u2, Rx), reads, cr2, a2, b2, sc=capi.default_scoring(min_dp_score=20))
        for k in KEYS_2D:
            assert np.array_equal(g[k], o[k]), (len(Lx), len(Rx), k, g[k][:10], o[k][:10])


@pytest.mark.gpu
def test_differential_fuzz_small():
    """60 rounds of tools/gpu_fuzz.py (random regions, scoring, N bases, missing flanks, junction
    indels; all 1D modes and both 2D modes against the oracle).  profiles/r01e_fuzz.txt holds a
    3000-round run."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location(
        "gpu_fuzz", os.path.join(os.path.dirname(__file__), "..", "tools", "gpu_fuzz.py"))
    fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
    rng = np.random.default_rng(20260104)
    for _ in range(60):
        assert fz.fuzz_1d(rng) is None
        assert fz.fuzz_1d_multi(rng) is None
        assert fz.fuzz_2d(rng) is None


def test_concurrent_calls_from_two_threads(capi, oracle):
    """The C ABI is thread-safe per call: thread-local error text and allocation arena, pooled
    streams/events behind a mutex.  Two threads issue 1D, 2D and pair calls at the same time."""
    import threading
    d1 = synth.make_1d(16, "TATTG", (8, 30), "ont", kwin=(0, 42), anchor=300, seed=501)
    d2 = synth.make_1d(16, "CAG", (5, 41), "ont_q20", kwin=(0, 50), anchor=250, seed=502)
    j = synth.make_joint(6, alleles=((6, 4), (11, 3)), read_len=500, read_sd=30, anchor=300, seed=503)
    cr, k1, k2 = _cells(j, step=2)
    want = [oracle.round3_1d(d["regions"], d["reads"], d["kmin"], d["kmax"]) for d in (d1, d2)]
    want2d = oracle.joint_2d(j["region"], j["reads"], cr, k1, k2)
    errors = []

    def worker(which):
        try:
            for rep in range(6):
                d, o = ((d1, want[0]), (d2, want[1]))[(which + rep) % 2]
                g = capi.round3_1d(d["regions"], d["reads"], d["kmin"], d["kmax"])
                for k in KEYS_1D:
                    assert np.array_equal(g[k], o[k]), (which, rep, k)
                g2 = capi.joint_2d(j["region"], j["reads"], cr, k1, k2)
                for k in KEYS_2D:
                    assert np.array_equal(g2[k], want2d[k]), (which, rep, k)
                with pytest.raises(capi.NraError) as e:          # the error text belongs to this thread's call
                    capi.round3_1d(d["regions"], ["A" * (200001 + which)], [0], [1])
                assert str(200001 + which) in str(e.value)
        except BaseException as exc:                              # noqa: BLE001 - reported in the main thread
            errors.append(exc)

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_1d_largest_single_block_reads(capi, oracle):
    """Reads of 3000-3072 bases: the largest rows-per-lane instantiation (R = 48) and the widest score
    range the doubled (origin-bit) int16 cells see with the default scoring."""
    rng = np.random.default_rng(47)
    L, R, u = synth.rand_seq(rng, 150), synth.rand_seq(rng, 150), "CAG"
    reads, kmin, kmax = [], [], []
    for k in (944, 943, 920, 944):                        # 120 + 3k + 120 = 3072, 3069, 3000, 3072 bases
        core = L[-120:] + u * k + R[:120]
        read = synth.apply_errors(rng, core, "hifi")[:3072] if len(reads) else core      # one exactly 3072, error-free
        assert 2900 < len(read) <= 3072
        reads.append(read); kmin.append(k - 3); kmax.append(k + 3)
    reads.append(L[-100:] + u * 1000)                     # 3100 > 3072: chained, in the same batch
    kmin.append(998); kmax.append(1001)
    g = capi.round3_1d([(L, u, R)], reads, kmin, kmax)
    o = oracle.round3_1d([(L, u, R)], reads, kmin, kmax)
    for k in KEYS_1D:
        assert np.array_equal(g[k], o[k]), k
    assert max(o["best_score"]) > 5000


def test_align_pairs_long_queries(capi, oracle):
    """nra_align_pairs with queries of 3.1-7.9 kb (chained row blocks) next to short ones, as the
    round-2 estimate of a long core needs (nanoRepeat_bam.py:362)."""
    rng = np.random.default_rng(55)
    L, u = synth.rand_seq(rng, 300), "GGCCCC"
    target = L + u * 1400
    seqs = [target]
    for k in (500, 900, 1290, 30):
        seqs.append(synth.apply_errors(rng, L[-100:] + u * k + synth.rand_seq(rng, 80), "ont_q20")[:7990])
    seqs.append("N" * 10 + seqs[1][10:3500])
    pq = list(range(1, len(seqs))); pt = [0] * len(pq)
    g = capi.align_pairs(seqs, pq, pt)
    o = oracle.align_pairs(seqs, pq, pt)
    assert max(len(s) for s in seqs[1:]) > 7000
    for k in ("score", "tstart", "tend"):
        assert np.array_equal(g[k], o[k]), (k, g[k], o[k])
    with pytest.raises(capi.NraError) as e:
        capi.align_pairs([target, "A" * 200001], [1], [0])
    assert e.value.code == -3

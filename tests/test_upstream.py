"""Rows upstream of the hot path: anchor logic, core trimming, rounds 1-2,
against fixtures captured from the reference's Python (tests/golden/ref_upstream.json)."""
import json
import os

import numpy as np
import pytest

from nanorepeat_amd import upstream as U, round3 as R3

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_upstream.json")
FIELDS = ("read_name", "full_read_len", "left_anchor_is_good", "right_anchor_is_good", "both_anchors_are_good",
          "core_seq_start_pos", "core_seq_end_pos", "mid_seq_start_pos", "mid_seq_end_pos",
          "dist_between_anchors", "strand", "left_buffer_len", "right_buffer_len")


@pytest.fixture(scope="module")
def golden():
    return json.load(open(GOLDEN))


def _region(left, unit, right):
    rr = R3.RepeatRegion()
    rr.left_anchor_seq, rr.repeat_unit_seq, rr.right_anchor_seq = left, unit, right
    rr.left_anchor_len, rr.right_anchor_len = len(left), len(right)
    return rr


def _hits_from_paf(lines):
    """PAF text -> AnchorHit records, grouped by consecutive qname like
    find_anchor_locations_from_paf (nanoRepeat_bam.py:238-258)."""
    groups = []
    for line in lines:
        c = line.split("\t")
        qlen, qs, qe = int(c[1]), int(c[2]), int(c[3])
        if c[4] == "-":
            qs, qe = qlen - qe, qlen - qs                       # paf.py:70-74
        AS = int([x for x in c[12:] if x.startswith("AS:i:")][0][5:])
        hit = U.AnchorHit(c[0], qlen, qs, qe, c[4], c[5], AS, int(c[10]), int(c[11]))
        if groups and groups[-1][0].qname == hit.qname:
            groups[-1].append(hit)
        else:
            groups.append([hit])
    return groups


def test_anchor_logic_matches_reference(golden):
    for case in golden["anchor_logic"]:
        rr = _region("A" * 1000, "TATTG", "C" * 1000)
        for g in _hits_from_paf(case["lines"]):
            U.find_anchor_locations_for1read(g, rr)
        assert list(rr.read_dict) == list(case["reads"])
        for n, want in case["reads"].items():
            got = rr.read_dict[n]
            for f in FIELDS:
                assert getattr(got, f) == want[f], (n, f)


def _run_flow(e, aligner=None, scorer=None):
    rr = _region(e["left"], e["unit"], e["right"])
    reads = {r["name"]: r["seq"] for r in e["reads"]}
    U.find_anchor_locations_in_reads("ont", rr, 4, region_reads=reads, aligner=aligner)
    U.make_core_seq(rr, reads)
    U.round1_and_round2_estimation("ont", rr, 4, aligner=aligner)
    R3.round3_estimation("ont", False, rr, 4, scorer=scorer)
    return rr


def _check_flow(e, rr):
    assert list(rr.read_dict) == list(e["read_dict"])
    assert rr.read_core_seq_dict == e["core"]
    for n, want in e["read_dict"].items():
        got = rr.read_dict[n]
        for f in FIELDS:
            assert getattr(got, f) == want[f], (n, f)
        for f in ("round1_repeat_size", "round2_repeat_size", "round3_repeat_size"):
            g = getattr(got, f)
            assert (None if g is None else float(g)) == want[f], (n, f)


def test_steps_1_to_3_match_reference_with_oracle(oracle, golden):
    """quantify1repeat_from_bam steps 1-3 (nanoRepeat_bam.py:656-679): the reference's own run
    (fixture) vs this repo's host logic with the CPU oracle as the aligner."""
    for e in golden["e2e"]:
        _check_flow(e, _run_flow(e, aligner=oracle.align_pairs, scorer=oracle.round3_1d))


@pytest.mark.gpu
def test_steps_1_to_3_match_reference_on_gpu(capi, golden):
    for e in golden["e2e"]:
        _check_flow(e, _run_flow(e))


@pytest.mark.gpu
def test_align_pairs_matches_oracle(capi, oracle):
    from nanorepeat_amd import synth
    rng = np.random.default_rng(12)
    seqs, pq, pt = [], [], []
    for i in range(12):
        t = synth.rand_seq(rng, int(rng.integers(50, 4000)))
        a = int(rng.integers(0, max(1, len(t) - 40)))
        q = synth.apply_errors(rng, t[a:a + int(rng.integers(30, 900))], "ont")
        if i % 4 == 0:
            q = synth.revcomp(q)
        seqs += [t, q]
        pq += [2 * i + 1, 2 * i + 1]; pt += [2 * i, (2 * i + 2) % 24]
    seqs += ["", "ACGTNNNN" * 30, synth.rand_seq(rng, 3072), synth.rand_seq(rng, 20000)]
    pq += [24, 25, 26, 25, 1]; pt += [0, 25, 27, 24, 1]
    for over in ({}, {"min_dp_score": 0}):
        g = capi.align_pairs(seqs, pq, pt, sc=capi.default_scoring(**over))
        o = oracle.align_pairs(seqs, pq, pt, sc=oracle.default_scoring(**over))
        for k in ("score", "tstart", "tend"):
            assert np.array_equal(g[k], o[k]), (k, g[k], o[k])
    with pytest.raises(capi.NraError):                       # queries: <= 200 000 bases (chained above 3072)
        capi.align_pairs(["A" * 200001, "ACGT"], [0], [1])
    with pytest.raises(capi.NraError):                       # the traceback keeps one register block
        capi.align_pairs_cigar([synth.rand_seq(rng, 3073), "ACGT"], [0], [1])


@pytest.mark.gpu
def test_align_pairs_long_queries_and_targets(capi, oracle):
    """Beyond the int32 cells: a 9.5 kb core as the query (score > 16 bits of the packed word's half ->
    chained rows), a 70 kb read as the target of a 1 kb anchor (tstart beyond 16 bits -> int64 cells),
    both at once, and an N-rich pair; all equal the oracle."""
    from nanorepeat_amd import synth
    rng = np.random.default_rng(31)
    core_t = synth.rand_seq(rng, 400) + "TATTG" * 1800 + synth.rand_seq(rng, 400)
    core = synth.apply_errors(rng, core_t[300:-300], "ont_q20")
    read70 = synth.rand_seq(rng, 70000)
    anchor = synth.apply_errors(rng, read70[66000:67000], "ont")
    long_q = synth.apply_errors(rng, read70[1000:21000], "hifi")            # 20 kb query, 70 kb target
    seqs = [core_t, core, read70, anchor, long_q, ("ACGTN" * 800), ("ACGTN" * 900)]
    pq, pt = [1, 3, 4, 5, 3], [0, 2, 2, 6, 0]
    g = capi.align_pairs(seqs, pq, pt)
    o = oracle.align_pairs(seqs, pq, pt)
    for k in ("score", "tstart", "tend"):
        assert np.array_equal(g[k], o[k]), (k, g[k], o[k])
    assert g["score"][0] > 16000 and g["tstart"][1] > 65535 and g["score"][2] > 32767


def test_round3_keeps_round2_size_for_cores_beyond_the_kernel_limits(oracle, monkeypatch):
    """A core longer than the C ABI takes (or a template beyond its column limit) does not fail the batch."""
    import numpy as np
    from nanorepeat_amd import round3 as R3, synth
    monkeypatch.setattr(R3, "MAX_CORE_LEN", 8000)
    rng = np.random.default_rng(8)
    rr = R3.RepeatRegion("chr1\t100\t160\tCAG")
    rr.left_anchor_seq, rr.right_anchor_seq = synth.rand_seq(rng, 200), synth.rand_seq(rng, 200)
    for name, k, r2 in (("ok", 20, 19.6), ("huge", 2700, 2700.0)):
        rd = R3.Read(name, r2)
        rr.read_dict[name] = rd
        rr.read_core_seq_dict[name] = rr.left_anchor_seq[-100:] + "CAG" * k + rr.right_anchor_seq[:100]
    R3.round3_estimation("ont", False, rr, 1, scorer=oracle.round3_1d)
    assert rr.read_dict["ok"].round3_repeat_size == 20.0 and rr.read_dict["ok"].round3_status == 0
    assert rr.read_dict["huge"].round3_repeat_size == 2700.0 and rr.read_dict["huge"].round3_status == R3.READ_TOO_LONG


def test_many_region_calls_equal_per_region_calls(oracle):
    """find_anchor_locations_in_reads_many / round1_and_round2_estimation_many (one aligner call for
    all regions, chunked by bases) leave every region exactly as the per-region functions do."""
    import copy
    from nanorepeat_amd import round3 as R3, synth
    rng = np.random.default_rng(17)
    regions, reads_by_region = [], []
    for g, unit in enumerate(("CAG", "TATTG", "AT", "GGCCCC")):
        left, right = synth.rand_seq(rng, 400), synth.rand_seq(rng, 400)
        rr = R3.RepeatRegion(f"chr2\t{1000 * g}\t{1000 * g + 30}\t{unit}")
        rr.left_anchor_seq, rr.right_anchor_seq, rr.left_anchor_len, rr.right_anchor_len = left, right, 400, 400
        reads = {}
        for i in range(0 if g == 2 else 7):                       # one region without reads
            k = (9, 24)[i % 2]
            s = synth.apply_errors(rng, left[-250:] + unit * k + right[:250], "ont_q20")
            reads[f"g{g}r{i}"] = synth.revcomp(s) if i % 3 == 1 else s
        reads["junk%d" % g] = synth.rand_seq(rng, 300)
        regions.append(rr); reads_by_region.append(reads)
    one, many = copy.deepcopy(regions), copy.deepcopy(regions)
    for rr, reads in zip(one, reads_by_region):
        U.find_anchor_locations_in_reads("ont_q20", rr, 1, region_reads=reads, aligner=oracle.align_pairs)
        U.make_core_seq(rr, reads)
        U.round1_and_round2_estimation("ont_q20", rr, 1, aligner=oracle.align_pairs)
    for limit in (U.MAX_BASES_PER_CALL, 3000):                     # 3000: a chunk per region
        test = copy.deepcopy(many)
        sizes = [2 * sum(len(s) for s in r.values()) for r in reads_by_region]
        assert len(U._chunks_by_bases(sizes, limit)) == (1 if limit > 10 ** 6 else 4)
        U.find_anchor_locations_in_reads_many("ont_q20", test, reads_by_region, aligner=oracle.align_pairs, max_bases=limit)
        for rr, reads in zip(test, reads_by_region):
            U.make_core_seq(rr, reads)
        U.round1_and_round2_estimation_many("ont_q20", test, aligner=oracle.align_pairs, max_bases=limit)
        for a, b in zip(one, test):
            assert list(a.read_dict) == list(b.read_dict) and a.read_core_seq_dict == b.read_core_seq_dict
            for n in a.read_dict:
                ra, rb = a.read_dict[n], b.read_dict[n]
                assert ((ra.round1_repeat_size, ra.round2_repeat_size, ra.strand, ra.core_seq_start_pos, ra.core_seq_end_pos) ==
                        (rb.round1_repeat_size, rb.round2_repeat_size, rb.strand, rb.core_seq_start_pos, rb.core_seq_end_pos))


def test_over_long_reads_and_cores_are_left_out_not_fatal(oracle, monkeypatch, capsys):
    """one read beyond the aligner's target limit, or one core beyond its query limit, must not
    abort every region: it is skipped, recorded on its region and reported; the other reads are unaffected."""
    import copy
    from nanorepeat_amd import pipeline, round3 as R3, synth
    rng = np.random.default_rng(23)
    regions, reads_by_region = [], []
    for g, unit in enumerate(("CAG", "TATTG")):
        left, right = synth.rand_seq(rng, 300), synth.rand_seq(rng, 300)
        rr = R3.RepeatRegion(f"chr3\t{500 * g}\t{500 * g + 30}\t{unit}")
        rr.left_anchor_seq, rr.right_anchor_seq, rr.left_anchor_len, rr.right_anchor_len = left, right, 300, 300
        reads = {f"g{g}r{i}": synth.apply_errors(rng, left[-200:] + unit * (8 + 5 * (i % 2)) + right[:200], "ont_q20")
                 for i in range(5)}
        regions.append(rr); reads_by_region.append(reads)
    base = copy.deepcopy(regions)
    pipeline.quantify_regions(base, reads_by_region, aligner=oracle.align_pairs, scorer=oracle.round3_1d)
    # the same with a huge read in region 0 and a read with a huge core in region 1
    monkeypatch.setattr(U, "MAX_READ_LEN", 5000)
    monkeypatch.setattr(U, "MAX_CORE_LEN", 1500)
    reads_by_region[0]["giant"] = synth.rand_seq(rng, 6000)
    l1, r1 = regions[1].left_anchor_seq, regions[1].right_anchor_seq
    reads_by_region[1]["expanded"] = l1[-200:] + "TATTG" * 400 + r1[:200]                 # core of 2200 bases
    test = copy.deepcopy(regions)
    pipeline.quantify_regions(test, reads_by_region, aligner=oracle.align_pairs, scorer=oracle.round3_1d)
    assert "giant" in test[0].skipped_reads and "expanded" in test[1].skipped_reads
    assert "giant" not in test[0].read_dict and test[1].read_dict["expanded"].round3_repeat_size is None
    for a, b in zip(base, test):
        for n, ra in a.read_dict.items():
            assert b.read_dict[n].round3_repeat_size == ra.round3_repeat_size
    err = capsys.readouterr().err
    assert err.count("NOTICE") == 2 and "giant" in err and "expanded" in err

"""Joint round 1 (nanoRepeat_joint.py:509-649) against fixtures of the reference's own
run (tests/golden/make_golden.py joint_round1: the oracle answers its aligner calls)."""
import json
import os

import pytest

from nanorepeat_amd import joint
from nanorepeat_amd.paf import PAF

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def fx():
    with open(os.path.join(HERE, "golden", "ref_joint_round1.json")) as f:
        return json.load(f)


def _repeats(case):
    r1 = joint.Repeat().init_from_string(case["repeat1"]); r2 = joint.Repeat().init_from_string(case["repeat2"])
    r1.max_size += 10; r2.max_size += 10            # nanoRepeat_joint.py:200-201
    return r1, r2


def _check(est, case):
    assert {k: list(v) for k, v in est.repeat1_count_range_dict.items()} == case["repeat1_count_range"]
    assert {k: list(v) for k, v in est.repeat2_count_range_dict.items()} == case["repeat2_count_range"]
    assert {k: list(v) for k, v in est.potential_repeat_region_dict.items()} == case["potential_repeat_region"]
    assert sorted(est.bad_reads_set) == case["bad_reads"]


def test_round1_from_reference_paf(fx):
    for case in fx["cases"]:
        r1, r2 = _repeats(case)
        left, right, ll, rl = joint.round1_templates(case["chrom"], r1, r2, case["max_anchor_len"])
        pafs = [PAF(line.split("\t")) for line in case["round1_paf"].split("\n") if line.strip()]
        assert {p.tname for p in pafs} == {left[0], right[0]}
        _check(joint.round1_estimation_from_paf(pafs, r1, r2, ll, rl), case)


def _run(case, tmp_path, **kw):
    r1, r2 = _repeats(case)
    fastq = {n: f"@{n}\n{s}\n+\n{'I' * len(s)}\n" for n, s in case["reads"]}
    est = joint.initial_estimate_repeat_size(case["chrom"], fastq, "ont", 1, r1, r2, case["max_anchor_len"],
                                             out_dir=str(tmp_path), save_paf=True, **kw)
    return est, (tmp_path / "round1.paf").read_text()


def test_round1_with_oracle_aligners(fx, oracle, tmp_path):
    for case in fx["cases"]:
        est, paf_text = _run(case, tmp_path, aligner=oracle.align_pairs, cigar_aligner=oracle.align_pairs_cigar)
        _check(est, case)
        assert sorted(paf_text.split("\n")) == sorted(case["round1_paf"].split("\n"))


@pytest.mark.gpu
def test_round1_on_gpu(fx, capi, tmp_path):
    for case in fx["cases"]:
        est, paf_text = _run(case, tmp_path)
        _check(est, case)
        assert sorted(paf_text.split("\n")) == sorted(case["round1_paf"].split("\n"))


@pytest.mark.gpu
def test_cigar_chunking(capi):
    import numpy as np
    from nanorepeat_amd import synth
    rng = np.random.default_rng(4)
    seqs = [synth.rand_seq(rng, 300) for _ in range(6)]
    seqs += [synth.apply_errors(rng, s[40:260], "ont") for s in seqs[:6]]
    pq, pt = [6, 7, 8, 9, 10, 11, 6], [0, 1, 2, 3, 4, 5, 3]
    whole = capi.align_pairs_cigar(seqs, pq, pt)
    parts = capi.align_pairs_cigar_chunked(seqs, pq, pt, chunk_bytes=2 * 220 * 300 + 10)
    for k in whole:
        assert list(whole[k]) == list(parts[k]), k


def _joint_files(tmp_path, n=40, long_every=0):
    import numpy as np
    from nanorepeat_amd import synth
    rng = np.random.default_rng(12)
    left, right = synth.rand_seq(rng, 1100), synth.rand_seq(rng, 1100)
    mid = "CAACAGCCGCCAC"
    chrom = left + "CAG" * 19 + mid + "CCG" * 9 + right
    s1 = len(left); e1 = s1 + 57; s2 = e1 + len(mid); e2 = s2 + 27
    (tmp_path / "ref.fa").write_text(">chrX other\nACGT\n>chrJ x\n" + "\n".join(chrom[i:i + 70] for i in range(0, len(chrom), 70)) + "\n>chrZ\nGG\n")
    truth, lines = {}, []
    for i in range(n):
        a, b = ((17, 10), (55, 7))[i % 2]
        s = synth.apply_errors(rng, left[-400:] + "CAG" * a + mid + "CCG" * b + right[:400], "ont_q20")
        if long_every and i % long_every == 1:       # a whole-genome style read: kilobases around the locus
            s = synth.rand_seq(rng, 2600 + 100 * i) + s + synth.rand_seq(rng, 1900)
        if i % 3 == 0:
            s = synth.revcomp(s)
        lines.append(f"@jq{i:02d} x\n{s}\n+\n{'I' * len(s)}\n"); truth[f"jq{i:02d}"] = (a, b)
    (tmp_path / "reads.fastq").write_text("".join(lines))
    return truth, f"chrJ:{s1}:{e1}:CAG:200", f"chrJ:{s2}:{e2}:CCG:20"


def _check_joint_outputs(tmp_path, truth, est, alleles):
    assert set(est.repeat1_count_dict) == set(truth)
    close = [abs(est.repeat1_count_dict[n] - truth[n][0]) <= 1 and abs(est.repeat2_count_dict[n] - truth[n][1]) <= 1 for n in truth]
    assert sum(close) >= 0.85 * len(truth)
    got = [(a.repeat1_median_size, a.repeat2_median_size) for a in alleles]
    assert len(got) == 2 and all(abs(g[0] - w[0]) <= 1 and abs(g[1] - w[1]) <= 1 for g, w in zip(got, [(17, 10), (55, 7)])), got
    summary = (tmp_path / "out.summary.txt").read_text()
    assert "Method\t2D-GMM\nNum_Alleles\t2\n" in summary
    assert (tmp_path / "out.repeat_size.txt").read_text().count("\n") == len(truth) + 2
    assert (tmp_path / "out.phased_reads.txt").read_text().count("\n") == sum(a.num_reads for a in alleles) + 2
    assert (tmp_path / "out.allele1.fastq").exists() and (tmp_path / "out.allele2.fastq").exists()


def test_joint_command_from_files_with_oracle(oracle, tmp_path):
    from nanorepeat_amd import pipeline
    truth, rs1, rs2 = _joint_files(tmp_path, n=24)
    est, alleles = pipeline.quantify_joint(str(tmp_path / "reads.fastq"), str(tmp_path / "ref.fa"), rs2, rs1,
                                           str(tmp_path / "out"), seed=9, aligner=oracle.align_pairs,
                                           cigar_aligner=oracle.align_pairs_cigar, scorer=oracle.joint_2d)
    _check_joint_outputs(tmp_path, truth, est, alleles)


@pytest.mark.gpu
def test_joint_command_from_files_gpu_equals_oracle(capi, oracle, tmp_path):
    from nanorepeat_amd import pipeline
    truth, rs1, rs2 = _joint_files(tmp_path, n=24)
    est, alleles = pipeline.quantify_joint(str(tmp_path / "reads.fastq"), str(tmp_path / "ref.fa"), rs1, rs2,
                                           str(tmp_path / "out"), seed=9)
    _check_joint_outputs(tmp_path, truth, est, alleles)
    gpu_text = (tmp_path / "out.repeat_size.txt").read_text()
    (tmp_path / "o").mkdir()
    pipeline.quantify_joint(str(tmp_path / "reads.fastq"), str(tmp_path / "ref.fa"), rs1, rs2,
                            str(tmp_path / "o" / "out"), seed=9, aligner=oracle.align_pairs,
                            cigar_aligner=oracle.align_pairs_cigar, scorer=oracle.joint_2d)
    assert gpu_text == (tmp_path / "o" / "out.repeat_size.txt").read_text()


def test_joint_command_long_reads_with_oracle(oracle, tmp_path):
    """Reads longer than one register block (3072 bases): round 1 aligns them as DP targets, the grid
    rounds score the full read (nanoRepeat_joint.py:332,408) cell by cell in chained row blocks."""
    from nanorepeat_amd import pipeline
    truth, rs1, rs2 = _joint_files(tmp_path, n=12, long_every=3)
    est, alleles = pipeline.quantify_joint(str(tmp_path / "reads.fastq"), str(tmp_path / "ref.fa"), rs1, rs2,
                                           str(tmp_path / "out"), seed=9, aligner=oracle.align_pairs,
                                           cigar_aligner=oracle.align_pairs_cigar, scorer=oracle.joint_2d)
    _check_joint_outputs(tmp_path, truth, est, alleles)


@pytest.mark.gpu
def test_joint_command_long_reads_gpu_equals_oracle(capi, oracle, tmp_path):
    from nanorepeat_amd import pipeline
    truth, rs1, rs2 = _joint_files(tmp_path, n=12, long_every=3)
    est, alleles = pipeline.quantify_joint(str(tmp_path / "reads.fastq"), str(tmp_path / "ref.fa"), rs1, rs2,
                                           str(tmp_path / "out"), seed=9)
    _check_joint_outputs(tmp_path, truth, est, alleles)
    (tmp_path / "o").mkdir()
    pipeline.quantify_joint(str(tmp_path / "reads.fastq"), str(tmp_path / "ref.fa"), rs1, rs2,
                            str(tmp_path / "o" / "out"), seed=9, aligner=oracle.align_pairs,
                            cigar_aligner=oracle.align_pairs_cigar, scorer=oracle.joint_2d)
    assert (tmp_path / "out.repeat_size.txt").read_text() == (tmp_path / "o" / "out.repeat_size.txt").read_text()


def test_bulk_dict_lookups_of_the_grid_rounds():
    """joint._rows_with / _values_of: the read numbers every dict holds and their values, with all names
    present (one C-level pass) and with some missing (the reference skips those reads)."""
    import numpy as np
    from nanorepeat_amd import joint as J
    names = [f"r{i}" for i in range(7)]
    full = {n: (i, i + 5) for i, n in enumerate(names)}
    part = {n: float(i) for i, n in enumerate(names) if i not in (2, 5)}
    assert J._rows_with(names, full).tolist() == list(range(7))
    assert J._rows_with(names, full, part).tolist() == [0, 1, 3, 4, 6]
    assert J._rows_with(names, {}).tolist() == []
    rows = J._rows_with(names, full, part)
    assert J._values_of(full, names, rows, np.int64, 2).tolist() == [[i, i + 5] for i in (0, 1, 3, 4, 6)]
    assert J._values_of(part, names, rows, np.float64).tolist() == [0.0, 1.0, 3.0, 4.0, 6.0]
    assert J._values_of(full, names, np.arange(7), np.int64, 2).tolist() == [[i, i + 5] for i in range(7)]   # flat pass
    shuffled = {n: full[n] for n in reversed(names)}               # same names, another order: looked up by name
    assert J._values_of(shuffled, names, np.arange(7), np.int64, 2).tolist() == [[i, i + 5] for i in range(7)]
    assert J._values_of(full, names[:1], np.arange(1), np.int64, 2).tolist() == [[0, 5]]
    assert len(J._values_of(full, names, np.zeros(0, np.int64), np.int64, 2)) == 0
    bigger = dict(full); bigger["extra"] = (9, 9)                  # more keys than names: still the fast path
    assert J._rows_with(names, bigger).tolist() == list(range(7))

"""Ingestion helpers against fixtures from the reference's Python, and the
region pipeline (steps 1-3) from files."""
import json
import hashlib
import os

import numpy as np
import pytest

from nanorepeat_amd import io as IO, pipeline, round3 as R3, synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def golden():
    return json.load(open(os.path.join(GOLDEN, "ref_io.json")))


def test_bed_reader_matches_reference(golden, tmp_path):
    p = tmp_path / "r.bed"
    p.write_bytes(golden["bed"]["text"].encode())            # CRLF, no trailing newline
    regs = IO.read_repeat_region_file(str(p))
    got = [[r.chrom, r.start_pos, r.end_pos, r.repeat_unit_seq, r.to_unique_id()] for r in regs]
    assert got == golden["bed"]["regions"]
    with pytest.raises(ValueError):
        R3.RepeatRegion("chr1\t5\t9")


def test_fasta_reader_and_flanks_match_reference(golden, tmp_path):
    p = tmp_path / "ref.fa"
    p.write_text(golden["fasta"]["text"])
    d = IO.fasta_file2dict(str(p))
    assert list(d) == golden["fasta"]["names"]
    assert [len(v) for v in d.values()] == golden["fasta"]["lens"]
    assert [hashlib.sha1(v.encode()).hexdigest() for v in d.values()] == golden["fasta"]["sha1"]
    import gzip
    gz = tmp_path / "ref.fa.gz"
    with gzip.open(gz, "wt") as f:
        f.write(golden["fasta"]["text"])
    assert IO.fasta_file2dict(str(gz)) == d
    for c in golden["flanks"]:
        rr = R3.RepeatRegion(f"{c['chrom']}\t{c['start']}\t{c['end']}\tCAG")
        IO.extract_ref_sequence(d, rr, c["anchor_len"])
        assert (rr.left_anchor_seq, rr.right_anchor_seq, rr.mid_ref_seq) == (c["left"], c["right"], c["mid"])
        assert rr.anchor_len == c["anchor_len_after"]
    with pytest.raises(KeyError):
        IO.extract_ref_sequence(d, R3.RepeatRegion("chr9\t1\t5\tCAG"))
    with pytest.raises(ValueError):
        IO.extract_ref_sequence(d, R3.RepeatRegion("chr4\t3500\t3600\tCAG"))


def test_motif_check_and_edit_distance():
    assert IO.edit_distance("kitten", "sitting") == 3 and IO.edit_distance("", "abc") == 3
    rr = R3.RepeatRegion("c\t0\t12\tCAG"); rr.mid_ref_seq = "CAGCAGCAACAG"
    assert IO.check_repeat_motif_in_ref(rr)
    rr.mid_ref_seq = "ACGTTGCATGCA"
    assert not IO.check_repeat_motif_in_ref(rr) and rr.ref_has_issue


def _make_files(tmp_path, n_regions=3, reads_per_region=8):
    rng = np.random.default_rng(77)
    chrom = synth.rand_seq(rng, 600)
    bed, truth, fastqs = [], [], []
    for g in range(n_regions):
        unit = ["TATTG", "CAG", "AAGGG"][g % 3]
        k_ref = 6 + g
        start = len(chrom)
        chrom += unit * k_ref + synth.rand_seq(rng, 700)
        bed.append(f"chrS\t{start}\t{start + len(unit) * k_ref}\t{unit}")
    (tmp_path / "ref.fa").write_text(">chrS test\n" + "\n".join(chrom[i:i + 60] for i in range(0, len(chrom), 60)) + "\n")
    (tmp_path / "regions.bed").write_text("\r\n".join(bed))
    for g, line in enumerate(bed):
        _, st, en, unit = line.split("\t"); st, en = int(st), int(en)
        alleles = (7 + g, 19 + 2 * g)
        lines, kt = [], {}
        for i in range(reads_per_region):
            k = alleles[i % 2]
            s = chrom[st - 450:st] + unit * k + chrom[en:en + 450]
            s = synth.apply_errors(rng, s, "ont_q20")
            if i % 3 == 0:
                s = synth.revcomp(s)
            lines.append(f"@g{g}r{i}\n{s}\n+\n{'I' * len(s)}\n"); kt[f"g{g}r{i}"] = k
        (tmp_path / f"region{g}.fastq").write_text("".join(lines))
        truth.append(kt)
    return truth


def _run_pipeline(tmp_path, **kw):
    ref = IO.fasta_file2dict(str(tmp_path / "ref.fa"))
    regions = IO.read_repeat_region_file(str(tmp_path / "regions.bed"))
    reads = []
    for g, rr in enumerate(regions):
        IO.extract_ref_sequence(ref, rr, anchor_len=400)
        assert IO.check_repeat_motif_in_ref(rr)
        reads.append(IO.read_fastq(str(tmp_path / f"region{g}.fastq")))
    texts = pipeline.quantify_regions(regions, reads, "ont_q20", **kw)
    return regions, texts


def test_pipeline_from_files_with_oracle(oracle, tmp_path):
    truth = _make_files(tmp_path)
    regions, texts = _run_pipeline(tmp_path, aligner=oracle.align_pairs, scorer=oracle.round3_1d)
    for rr, kt, text in zip(regions, truth, texts):
        assert len(rr.read_dict) == len(kt)
        sizes = {l.split("\t")[0]: float(l.split("\t")[1]) for l in text.strip().split("\n")[2:]}
        assert set(sizes) == set(kt)
        assert np.mean([abs(sizes[n] - kt[n]) <= 1 for n in kt]) >= 0.85
        assert text.startswith(f"##Repeat_Region={rr.to_unique_id()}\n#Read_Name\tRepeat_Size\n")


@pytest.mark.gpu
def test_pipeline_from_files_gpu_equals_oracle(capi, oracle, tmp_path):
    _make_files(tmp_path)
    _, t_gpu = _run_pipeline(tmp_path)
    _, t_cpu = _run_pipeline(tmp_path, aligner=oracle.align_pairs, scorer=oracle.round3_1d)
    assert t_gpu == t_cpu


def test_pipeline_step4_phasing_and_final_table(oracle, tmp_path):
    """Steps 1-4 from files: the two alleles of every region come out of the mixture and the
    final table has the reference's row layout (repeat_region.py:186-191)."""
    _make_files(tmp_path, n_regions=2, reads_per_region=24)
    regions, _ = _run_pipeline(tmp_path, aligner=oracle.align_pairs, scorer=oracle.round3_1d)
    for g, rr in enumerate(regions):
        rr.out_prefix = str(tmp_path / f"out{g}")
        rr.region_fq_file = str(tmp_path / f"region{g}.fastq")
    rows = pipeline.phase_regions(regions, "ont_q20", seed=3, out_tsv_file=str(tmp_path / "x.NanoRepeat_output.tsv"))
    assert (tmp_path / "x.NanoRepeat_output.tsv").read_text() == "".join(rows)
    for g, (rr, row) in enumerate(zip(regions, rows)):
        cols = row.rstrip("\n").split("\t")
        assert cols[0] == "chrS" and cols[3] == rr.repeat_unit_seq and len(cols) == 9
        assert int(cols[4]) == 2 and {int(cols[5]), int(cols[6])} == {7 + g, 19 + 2 * g}
        assert cols[7].startswith("Allele_Repeat_Size;Allele_Num_Support_Reads|")
        assert cols[8].count("|") == len(rr.read_dict)
        assert (tmp_path / f"out{g}.summary.txt").read_text().count("Num_Alleles=2") == 1
        n_fq = sum((tmp_path / f"out{g}.allele{a}.fastq").read_text().count("\n+\n") for a in (1, 2))
        assert 0 < n_fq <= len(rr.read_dict)


def test_joint_mode_readers_match_reference(golden, tmp_path):
    import hashlib
    fa = tmp_path / "ref.fa"; fa.write_text(golden["fasta"]["text"])
    for name, sha in golden["one_chr"].items():
        assert hashlib.sha1(IO.read_one_chr_from_fasta_file(str(fa), name).encode()).hexdigest() == sha, name
    fq = tmp_path / "x.fastq"; fq.write_text(golden["fastq_dict"]["text"])
    got = IO.fastq_file_to_dict(str(fq))
    assert got == golden["fastq_dict"]["dict"] and list(got) == list(golden["fastq_dict"]["dict"])


@pytest.mark.gpu
def test_pipeline_with_a_long_expansion_gpu_equals_oracle(capi, oracle):
    """An allele of 1150 CAG units (core of 3.6 kb: more than one register block) goes through
    anchors -> core -> round 2 (chained pair alignment) -> round 3 (chained sweeps) like the short one."""
    rng = np.random.default_rng(91)
    left, right = synth.rand_seq(rng, 500), synth.rand_seq(rng, 500)
    out = {}
    for name, kw in (("gpu", {}), ("cpu", dict(aligner=oracle.align_pairs, scorer=oracle.round3_1d))):
        rr = R3.RepeatRegion("chrL\t500\t560\tCAG")
        rr.left_anchor_seq, rr.right_anchor_seq = left, right
        rr.left_anchor_len = rr.right_anchor_len = 500
        rng2 = np.random.default_rng(92)
        reads = {}
        for i, k in enumerate((30, 1150, 31, 1148)):
            s = synth.apply_errors(rng2, left[-300:] + "CAG" * k + right[:300], "hifi")
            reads[f"x{i}"] = synth.revcomp(s) if i == 2 else s
        pipeline.quantify_regions([rr], [reads], "hifi", **kw)
        out[name] = {n: (r.round2_repeat_size, r.round3_repeat_size, r.round3_status) for n, r in rr.read_dict.items()}
    assert out["gpu"] == out["cpu"]
    sizes = [out["gpu"][f"x{i}"][1] for i in range(4)]
    assert abs(sizes[0] - 30) <= 1 and abs(sizes[1] - 1150) <= 12 and abs(sizes[3] - 1148) <= 12, sizes   # ~1 % indel noise


@pytest.mark.gpu
def test_pipeline_70kb_read_and_9kb_core_in_a_multi_region_batch(capi, oracle):
    """whole-genome ONT reads over 65 kb and cores over 8 kb are routine and must not abort a
    batch.  Region 0 holds a 70 kb read (the anchors' DP target: int64 extents), region 1 a read whose
    core is 9.2 kb (round 2: chained pair alignment; round 3: chained int32 sweeps with the reference's
    window for r2 = 1800, K = 181); both regions also hold ordinary reads.  GPU == oracle, nothing skipped."""
    rng = np.random.default_rng(131)
    regions, reads_by_region = [], []
    for g, unit in enumerate(("CAG", "TATTG")):
        left, right = synth.rand_seq(rng, 600), synth.rand_seq(rng, 600)
        rr = R3.RepeatRegion(f"chrW\t{3000 * g + 600}\t{3000 * g + 660}\t{unit}")
        rr.left_anchor_seq, rr.right_anchor_seq, rr.left_anchor_len, rr.right_anchor_len = left, right, 600, 600
        reads = {f"g{g}r{i}": synth.apply_errors(rng, left[-400:] + unit * (11 + 7 * (i % 2)) + right[:400], "ont_q20")
                 for i in range(4)}
        if g == 0:
            body = left + unit * 25 + right
            reads["long70k"] = synth.rand_seq(rng, 41000) + synth.apply_errors(rng, body, "ont_q20") + synth.rand_seq(rng, 28000)
        else:
            reads["core9k"] = synth.apply_errors(rng, left[-500:] + unit * 1800 + right[:500], "hifi")
        regions.append(rr); reads_by_region.append(reads)
    assert len(reads_by_region[0]["long70k"]) > 70000
    import copy
    out = {}
    for name, kw in (("gpu", {}), ("cpu", dict(aligner=oracle.align_pairs, scorer=oracle.round3_1d))):
        test = copy.deepcopy(regions)
        pipeline.quantify_regions(test, reads_by_region, "ont_q20", **kw)
        out[name] = [{n: (r.strand, r.core_seq_start_pos, r.core_seq_end_pos, r.round2_repeat_size, r.round3_repeat_size,
                          r.round3_status) for n, r in rr.read_dict.items()} for rr in test]
        assert not any(getattr(rr, "skipped_reads", None) for rr in test)
    assert out["gpu"] == out["cpu"]
    long_read, core = out["gpu"][0]["long70k"], out["gpu"][1]["core9k"]
    assert long_read[1] > 40000 and abs(long_read[4] - 25) <= 1                # placed beyond 16-bit extents, sized
    assert core[2] - core[1] > 9000 and abs(core[4] - 1800) <= 20

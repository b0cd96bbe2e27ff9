"""BAM ingestion (nanoRepeat_bam.py:576-600): the FASTQ text is pinned by the
reference's own function run with a stand-in for pysam (tests/golden/ref_io.json, bam_extract);
the container reader is checked on BAM / BAI files written here with the standard library."""
import json
import os
import struct
import sys
import types
import zlib

import pytest

from nanorepeat_amd import bam as B
from nanorepeat_amd.round3 import RepeatRegion

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def fx():
    with open(os.path.join(HERE, "golden", "ref_io.json")) as f:
        return json.load(f)["bam_extract"]


# ---------------------------------------------------------------------------- a tiny BAM writer
def _bgzf_block(payload):
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    data = c.compress(payload) + c.flush()
    bsize = 12 + 6 + len(data) + 8 - 1
    return (b"\x1f\x8b\x08\x04" + b"\x00" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize) +
            data + struct.pack("<II", zlib.crc32(payload), len(payload)))


def _record(name, seq, quals, tid, pos, ref_len, flag=0):
    seq = seq or ""
    cigar = [(ref_len << 4) | 0] if ref_len and seq else []          # one M run (its length need not match l_seq here)
    codes = [B._SEQ_CODES.index(c) for c in seq] + [0]
    packed = bytes((codes[i] << 4) | codes[i + 1] for i in range(0, len(seq), 2))
    q = bytes(quals) if quals is not None else b"\xff" * len(seq)
    body = struct.pack("<iiBBHHHiiii", tid, pos, len(name) + 1, 60, 0, len(cigar), flag, len(seq), -1, -1, 0)
    body += name.encode() + b"\x00" + b"".join(struct.pack("<I", c) for c in cigar) + packed + q
    return struct.pack("<i", len(body)) + body


def write_bam(path, refs, records, block=220, index=True):
    """records: (name, seq, quals, ref name, pos, end) sorted by (ref, pos).  Returns nothing; writes
    `path` and, with index=True, `path`.bai holding only the linear index (n_bin = 0)."""
    text = b"@HD\tVN:1.6\tSO:coordinate\n" + b"".join(f"@SQ\tSN:{n}\tLN:{l}\n".encode() for n, l in refs)
    head = b"BAM\x01" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(refs))
    for n, l in refs:
        head += struct.pack("<i", len(n) + 1) + n.encode() + b"\x00" + struct.pack("<i", l)
    names = [n for n, _ in refs]
    chunks, starts, upos = [head], [], len(head)
    for name, seq, quals, ref, pos, end in records:
        starts.append(upos)
        rec = _record(name, seq, quals, names.index(ref), pos, end - pos)
        chunks.append(rec); upos += len(rec)
    raw = b"".join(chunks)
    blocks, block_at = [], {}                # uncompressed offset of each block start -> compressed offset
    coff = 0
    for u in range(0, len(raw), block):
        block_at[u] = coff
        bl = _bgzf_block(raw[u:u + block]); blocks.append(bl); coff += len(bl)
    blocks.append(_bgzf_block(b""))          # EOF marker
    with open(path, "wb") as f:
        f.write(b"".join(blocks))
    if not index:
        return
    linear = [[] for _ in refs]
    for (name, seq, quals, ref, pos, end), u in zip(records, starts):
        v = (block_at[u - u % block] << 16) | (u % block)
        iv = linear[names.index(ref)]
        for w in range(pos >> 14, (max(end, pos + 1) - 1 >> 14) + 1):
            while len(iv) <= w:
                iv.append(0)
            if iv[w] == 0:
                iv[w] = v
    with open(path + ".bai", "wb") as f:
        f.write(b"BAI\x01" + struct.pack("<i", len(refs)))
        for iv in linear:
            f.write(struct.pack("<i", 0) + struct.pack("<i", len(iv)) + b"".join(struct.pack("<Q", v) for v in iv))


def _fastq_records(text):
    lines = text.split("\n")
    return [tuple(lines[i:i + 4]) for i in range(0, len(lines) - 1, 4)]


# ---------------------------------------------------------------------------- tests
def test_fastq_text_matches_reference_through_a_pysam_stand_in(fx, tmp_path, monkeypatch):
    calls = []

    class Read:
        def __init__(self, r):
            self.query_name, self.query_sequence, self.query_qualities, self.pos, self.end = r

    class AlignmentFile:
        def __init__(self, path, mode, reference_filename=None):
            calls.append(["open", os.path.basename(path), mode, reference_filename])

        def fetch(self, chrom, start, end):
            calls.append(["fetch", chrom, start, end])
            return [Read(r) for r in fx["records"] if r[3] < end and r[4] > start]

        def close(self):
            calls.append(["close"])

    monkeypatch.setitem(sys.modules, "pysam", types.SimpleNamespace(AlignmentFile=AlignmentFile))
    for case in fx["cases"]:
        chrom, st, en = case["region"]
        del calls[:]
        out = tmp_path / "o.fastq"
        n = B.extract_fastq_from_bam(str(tmp_path / "in.bam"), RepeatRegion(f"{chrom}\t{st}\t{en}\tCAG"), case["flank"],
                                     str(out), ref_fasta="ref.fa")
        assert out.read_text() == case["fastq"] and calls == case["calls"]
        assert n == case["fastq"].count("\n+\n")


@pytest.mark.parametrize("index", [True, False])
def test_container_reader_gives_the_same_reads(fx, tmp_path, monkeypatch, index):
    monkeypatch.setitem(sys.modules, "pysam", None)                 # import pysam -> ImportError
    recs = sorted(([n, s, q, "chr4", p, e] for n, s, q, p, e in fx["records"]), key=lambda r: r[4])
    recs = [["u0", "ACGT", [5] * 4, "chr1", 10, 900]] + recs + [["z9", "ACGT", None, "chrX", 5, 50]]
    path = str(tmp_path / "t.bam")
    write_bam(path, [("chr1", 5000), ("chr4", 200000), ("chrX", 1000)], recs, index=index)
    for case in fx["cases"]:
        chrom, st, en = case["region"]
        out = tmp_path / "o.fastq"
        B.extract_fastq_from_bam(path, RepeatRegion(f"{chrom}\t{st}\t{en}\tCAG"), case["flank"], str(out))
        got = _fastq_records(out.read_text())
        assert sorted(got) == sorted(_fastq_records(case["fastq"]))
    with pytest.raises(ValueError):
        B.extract_fastq_from_bam(path, RepeatRegion("chr9\t5\t9\tCAG"), 0, str(tmp_path / "x.fastq"))
    with pytest.raises(RuntimeError):
        B.open_alignment_file(str(tmp_path / "t.cram"))


def test_linear_index_windows_and_block_boundaries(tmp_path, monkeypatch):
    """Reads spread over many 16 kb windows (some empty), records straddling BGZF blocks, a read that
    spans several windows, an unmapped-with-position record."""
    monkeypatch.setitem(sys.modules, "pysam", None)
    import random
    rng = random.Random(3)
    recs = []
    for i, pos in enumerate(sorted(rng.sample(range(0, 150000), 60))):
        ln = rng.choice([30, 200, 700])
        seq = "".join(rng.choice("ACGTN") for _ in range(ln))
        recs.append([f"q{i:02d}", seq, [rng.randrange(0, 60) for _ in range(ln)] if i % 4 else None, "c", pos, pos + ln])
    recs.append(["long", "ACGTACGT", None, "c", 20000, 90000])
    recs.append(["far", "AC", None, "c", 400000, 400002])           # after empty windows
    recs.sort(key=lambda r: r[4])
    path = str(tmp_path / "w.bam")
    write_bam(path, [("c", 500000)], recs, block=157)
    nidx = str(tmp_path / "n.bam")
    write_bam(nidx, [("c", 500000)], recs, block=4000, index=False)
    for st, en in ((0, 100), (16384, 16385), (50000, 70000), (149000, 160000), (300000, 300100), (399990, 400001), (450000, 460000)):
        want = [(r[0], r[1]) for r in recs if r[4] < en and r[5] > st]
        for p in (path, nidx):
            f = B.BamFile(p)
            got = [(r.query_name, r.query_sequence) for r in f.fetch("c", st, en)]
            f.close()
            assert got == want, (st, en, p)
    f = B.BamFile(path)
    assert f.references == ["c"] and f.lengths == [500000]
    r = next(f.fetch("c", 0, 500000))
    assert r.query_qualities == recs[0][2] or recs[0][2] is None
    f.close()


def _bam_case(tmp_path):
    import numpy as np
    from nanorepeat_amd import synth
    rng = np.random.default_rng(5)
    chrom = synth.rand_seq(rng, 1500)
    s1 = len(chrom); chrom += "CAG" * 12; e1 = len(chrom); chrom += synth.rand_seq(rng, 1400)
    s2 = len(chrom); chrom += "TATTG" * 8; e2 = len(chrom); chrom += synth.rand_seq(rng, 1500)
    s3 = len(chrom); chrom += synth.rand_seq(rng, 30); e3 = len(chrom); chrom += synth.rand_seq(rng, 900)   # not a repeat in the reference
    (tmp_path / "ref.fa").write_text(">7 the reference names it without the prefix\n" + "\n".join(chrom[i:i + 80] for i in range(0, len(chrom), 80)) + "\n")
    (tmp_path / "r.bed").write_text(f"chr7\t{s1}\t{e1}\tCAG\nchr7\t{s2}\t{e2}\tTATTG\nchr7\t{s3}\t{e3}\tGGCCCC\n")
    recs, truth = [], {}
    for g, (st, en, unit, alleles) in enumerate(((s1, e1, "CAG", (9, 31)), (s2, e2, "TATTG", (6, 17)))):
        for i in range(20):
            k = alleles[i % 2]
            lo = st - 500 - 7 * i
            s = synth.apply_errors(rng, chrom[lo:st] + unit * k + chrom[en:en + 520], "ont_q20")
            name = f"g{g}r{i:02d}"
            if i % 3 == 0:
                s = synth.revcomp(s)                                    # stored as aligned: either strand occurs
            recs.append([name, s, [20] * len(s), "chr7", lo, en + 520])
            truth[name] = k
    recs.sort(key=lambda r: r[4])
    write_bam(str(tmp_path / "in.bam"), [("chr7", len(chrom))], recs, block=3000)
    return truth


def test_bam_command_from_files_with_oracle(oracle, tmp_path, monkeypatch):
    from nanorepeat_amd import pipeline
    monkeypatch.setitem(sys.modules, "pysam", None)
    truth = _bam_case(tmp_path)
    regions = pipeline.quantify_from_bam(str(tmp_path / "in.bam"), str(tmp_path / "ref.fa"), str(tmp_path / "r.bed"),
                                         str(tmp_path / "out"), data_type="ont_q20", anchor_len=400, seed=1,
                                         aligner=oracle.align_pairs, scorer=oracle.round3_1d)
    rows = (tmp_path / "out.NanoRepeat_output.tsv").read_text().split("\n")[:-1]
    assert len(rows) == 3
    for row, want in zip(rows, ((9, 31), (6, 17))):
        cols = row.split("\t")
        assert int(cols[4]) == 2 and sorted((int(cols[5]), int(cols[6]))) == sorted(want), cols[:8]
    assert rows[2].split("\t")[4] == "0"                               # fails the motif check: no alleles
    r0 = regions[0]
    sizes = {n: r.round3_repeat_size for n, r in r0.read_dict.items()}
    assert len(sizes) == 20 and sum(abs(sizes[n] - truth[n]) <= 1 for n in sizes) >= 18
    assert os.path.exists(r0.out_prefix + ".repeat_size.txt") and os.path.exists(r0.out_prefix + ".allele2.fastq")
    assert "/out.details/chr7/chr7-" in r0.out_prefix


@pytest.mark.gpu
def test_bam_command_from_files_gpu_equals_oracle(capi, oracle, tmp_path, monkeypatch):
    from nanorepeat_amd import pipeline
    monkeypatch.setitem(sys.modules, "pysam", None)
    _bam_case(tmp_path)
    args = (str(tmp_path / "in.bam"), str(tmp_path / "ref.fa"), str(tmp_path / "r.bed"))
    pipeline.quantify_from_bam(*args, str(tmp_path / "gpu"), data_type="ont_q20", anchor_len=400, seed=1)
    pipeline.quantify_from_bam(*args, str(tmp_path / "cpu"), data_type="ont_q20", anchor_len=400, seed=1,
                               aligner=oracle.align_pairs, scorer=oracle.round3_1d)
    assert (tmp_path / "gpu.NanoRepeat_output.tsv").read_text() == (tmp_path / "cpu.NanoRepeat_output.tsv").read_text()

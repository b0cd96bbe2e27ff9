"""Reads of a region from a coordinate-sorted BAM file -> FASTQ (SURVEY.md 8f-3;
nanoRepeat_bam.py:576-600).

The reference uses pysam for this one step.  `extract_fastq_from_bam` uses pysam too when it
can be imported; otherwise a small reader of the BAM container does the same job with the
standard library: BGZF blocks (gzip members with their size in the extra field), BAM records,
and the linear index of a `.bai` file to start near the region (without an index the file is
scanned from the start).  CRAM needs pysam.
"""
import os
import struct
import zlib

_SEQ_CODES = "=ACMGRSVTWYHKDBN"
_CIGAR_REF = (1, 0, 1, 1, 0, 0, 0, 1, 1)      # M I D N S H P = X: consumes the reference?


class BamRecord:
    """The fields `extract_fastq_from_bam` reads, named as pysam names them."""

    __slots__ = ("query_name", "query_sequence", "query_qualities", "reference_id", "reference_start",
                 "reference_end", "flag")

    def __init__(self, query_name, query_sequence, query_qualities, reference_id, reference_start, reference_end, flag):
        self.query_name, self.query_sequence, self.query_qualities = query_name, query_sequence, query_qualities
        self.reference_id, self.reference_start, self.reference_end, self.flag = reference_id, reference_start, reference_end, flag


class BgzfReader:
    """Sequential reader over BGZF blocks, positioned with virtual offsets (coffset << 16 | uoffset)."""

    def __init__(self, path):
        self._f = open(path, "rb")
        self._buf = b""
        self._pos = 0

    def close(self):
        self._f.close()

    def seek_virtual(self, voffset):
        self._f.seek(voffset >> 16)
        self._buf, self._pos = b"", 0
        self._fill()
        self._pos = voffset & 0xffff

    def _fill(self):
        """Next non-empty block into the buffer; False at end of file."""
        while True:
            head = self._f.read(12)
            if len(head) < 12:
                return False
            if head[0:4] != b"\x1f\x8b\x08\x04":
                raise ValueError("not a BGZF block (is this a BAM file?)")
            xlen = struct.unpack("<H", head[10:12])[0]
            extra = self._f.read(xlen)
            bsize, i = None, 0
            while i + 4 <= len(extra):
                si1, si2, slen = extra[i], extra[i + 1], struct.unpack("<H", extra[i + 2:i + 4])[0]
                if si1 == 66 and si2 == 67 and slen == 2:
                    bsize = struct.unpack("<H", extra[i + 4:i + 6])[0]
                i += 4 + slen
            if bsize is None:
                raise ValueError("BGZF block without a BC field")
            data = self._f.read(bsize + 1 - 12 - xlen - 8)
            tail = self._f.read(8)
            if len(tail) < 8:
                raise ValueError("truncated BGZF block")
            out = zlib.decompress(data, -15)
            if len(out) != struct.unpack("<I", tail[4:8])[0]:
                raise ValueError("BGZF block length mismatch")
            if out:
                self._buf, self._pos = out, 0
                return True

    def read(self, n):
        """Exactly n bytes, or fewer at end of file."""
        parts = []
        while n > 0:
            if self._pos >= len(self._buf) and not self._fill():
                break
            take = self._buf[self._pos:self._pos + n]
            self._pos += len(take)
            n -= len(take)
            parts.append(take)
        return b"".join(parts)


class BamFile:
    """Minimal stand-in for `pysam.AlignmentFile(path, "rb")`: `references`, `fetch`, `close`."""

    def __init__(self, path):
        self.path = path
        self._bgzf = BgzfReader(path)
        if self._bgzf.read(4) != b"BAM\x01":
            raise ValueError(f"{path}: not a BAM file")
        l_text = struct.unpack("<i", self._bgzf.read(4))[0]
        self.header_text = self._bgzf.read(l_text).split(b"\x00")[0].decode(errors="replace")
        n_ref = struct.unpack("<i", self._bgzf.read(4))[0]
        self.references, self.lengths = [], []
        for _ in range(n_ref):
            l_name = struct.unpack("<i", self._bgzf.read(4))[0]
            self.references.append(self._bgzf.read(l_name)[:-1].decode())
            self.lengths.append(struct.unpack("<i", self._bgzf.read(4))[0])
        self._linear = self._read_bai()
        self._first_record_known = False

    def close(self):
        self._bgzf.close()

    def _read_bai(self):
        """{ref_id: [virtual offsets of the 16 kb windows]} from `<bam>.bai` / `<stem>.bai`, or None."""
        for p in (self.path + ".bai", os.path.splitext(self.path)[0] + ".bai"):
            if os.path.exists(p):
                break
        else:
            return None
        with open(p, "rb") as f:
            d = f.read()
        if d[0:4] != b"BAI\x01":
            raise ValueError(f"{p}: not a BAI index")
        n_ref = struct.unpack_from("<i", d, 4)[0]
        off, linear = 8, {}
        for ref in range(n_ref):
            n_bin = struct.unpack_from("<i", d, off)[0]; off += 4
            for _ in range(n_bin):
                n_chunk = struct.unpack_from("<i", d, off + 4)[0]
                off += 8 + 16 * n_chunk
            n_intv = struct.unpack_from("<i", d, off)[0]; off += 4
            linear[ref] = list(struct.unpack_from(f"<{n_intv}Q", d, off)); off += 8 * n_intv
        return linear

    def _records(self):
        while True:
            head = self._bgzf.read(4)
            if len(head) < 4:
                return
            block = self._bgzf.read(struct.unpack("<i", head)[0])
            ref_id, pos, l_read_name, _mapq, _bin, n_cigar, flag, l_seq = struct.unpack_from("<iiBBHHHi", block, 0)
            p = 32
            name = block[p:p + l_read_name - 1].decode(); p += l_read_name
            ref_len = 0
            for v in struct.unpack_from(f"<{n_cigar}I", block, p):
                if _CIGAR_REF[v & 15 if (v & 15) < 9 else 1]:
                    ref_len += v >> 4
            p += 4 * n_cigar
            packed = block[p:p + (l_seq + 1) // 2]; p += (l_seq + 1) // 2
            seq = "".join(_SEQ_CODES[b >> 4] + _SEQ_CODES[b & 15] for b in packed)[:l_seq]
            qual = block[p:p + l_seq]
            quals = None if (l_seq == 0 or qual[0] == 0xff) else list(qual)
            yield BamRecord(name, seq if l_seq else None, quals, ref_id, pos, pos + max(ref_len, 1), flag)

    def fetch(self, contig, start, stop):
        """Records overlapping [start, stop) of `contig`, in file order (the file must be sorted)."""
        if contig not in self.references:
            raise ValueError(f"invalid contig `{contig}`")
        tid = self.references.index(contig)
        sorted_file = "SO:coordinate" in self.header_text.split("\n")[0] or self._linear is not None
        if self._linear is not None:
            iv = self._linear.get(tid, [])
            win = min(start >> 14, len(iv) - 1)
            voff = 0
            while win >= 0 and voff == 0:         # empty windows hold 0: back up to the previous one
                voff = iv[win] if iv else 0
                win -= 1
            if voff == 0:
                return
            self._bgzf.seek_virtual(voff)
        for rec in self._records():
            if sorted_file and (rec.reference_id > tid or rec.reference_id < 0 or
                                (rec.reference_id == tid and rec.reference_start >= stop)):
                return
            if rec.reference_id == tid and rec.reference_start < stop and rec.reference_end > start:
                yield rec


def open_alignment_file(in_bam_file, ref_fasta=None):
    """pysam's AlignmentFile when pysam is installed (BAM and CRAM), else the reader above (BAM)."""
    try:
        import pysam
        if hasattr(pysam, "AlignmentFile"):
            return pysam.AlignmentFile(in_bam_file, "rb", reference_filename=ref_fasta)
    except ImportError:
        pass
    if in_bam_file.lower().endswith(".cram"):
        raise RuntimeError("CRAM input needs pysam, which is not installed")
    return BamFile(in_bam_file)


def extract_fastq_from_bam(in_bam_file, repeat_region, flank_dist, out_fastq_file, ref_fasta=None):
    """nanoRepeat_bam.py:576-600: every read overlapping the region +- flank_dist, once (first
    record of a name wins), sequence as stored; missing qualities become '.' (Phred 13).
    Returns the number of reads written."""
    quality_shift = 33
    assert flank_dist >= 0
    start_pos = max(0, repeat_region.start_pos - flank_dist)
    end_pos = repeat_region.end_pos + flank_dist
    bam = open_alignment_file(in_bam_file, ref_fasta)
    written = set()
    try:
        with open(out_fastq_file, "w") as fastq:
            for read in bam.fetch(repeat_region.chrom, start_pos, end_pos):
                if not read.query_sequence or read.query_name in written:
                    continue
                if read.query_qualities is not None:
                    quals = "".join(chr(q + quality_shift) for q in read.query_qualities)
                else:
                    quals = chr(quality_shift + 13) * len(read.query_sequence)
                fastq.write(f"@{read.query_name}\n{read.query_sequence}\n+\n{quals}\n")
                written.add(read.query_name)
    finally:
        bam.close()
    return len(written)

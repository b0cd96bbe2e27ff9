"""PAF records: the wire format between the reference's aligner and its selectors
(paf.py:32-79), kept for compatibility artefacts -- the hot path itself
exchanges integers, not text."""
import re


class PAF:
    """Parses the 12 fixed columns plus AS:i / cg:Z / tp:A, and flips qstart/qend to the read's own
    strand on '-' records, exactly like the reference's class (paf.py:32-79)."""

    def __init__(self, col_list):
        if len(col_list) < 12:
            raise ValueError("number of columns should be >= 12 in a PAF file: " + "\t".join(col_list))
        self.qname, self.qlen, self.qstart, self.qend = col_list[0:4]
        self.strand = col_list[4]
        self.tname, self.tlen, self.tstart, self.tend = col_list[5:9]
        self.n_match, self.align_len, self.mapq = col_list[9:12]
        for f in ("qlen", "qstart", "qend", "tlen", "tstart", "tend", "n_match", "align_len", "mapq"):
            setattr(self, f, int(getattr(self, f)))
        self.is_primary = False
        self.align_score = -1
        self.cigar = ""
        for col in col_list[12:]:
            if col[0:5] == "AS:i:":
                self.align_score = int(col[5:])
            elif col[0:5] == "cg:Z:":
                self.cigar = col[5:]
            elif col == "tp:A:P":
                self.is_primary = True
            elif col == "tp:A:S":
                self.is_primary = False
        if self.strand not in "+-" or len(self.strand) != 1:
            raise ValueError(f"unknown strand: {self.strand}")
        if self.strand == "-":
            self.qstart, self.qend = self.qlen - self.qend, self.qlen - self.qstart


_CIGAR_ITEM = re.compile(r"(\d+)([=XIDNSHPM])")
_CIGAR_WHOLE = re.compile(r"(?:\d+[=XIDNSHPM])*\Z")


def cigar_ops(cigar):
    """[(op, length)] of a CIGAR string (tk.analysis_cigar_string, tk.py:379-401)."""
    if not _CIGAR_WHOLE.match(cigar):
        raise ValueError(f"unknown CIGAR operation in: {cigar[:40]}")
    return [(op, int(n)) for n, op in _CIGAR_ITEM.findall(cigar)]


def cigar_counts(cigar):
    """(matches, alignment block length) of an --eqx CIGAR."""
    n_match = block = 0
    for op, n in cigar_ops(cigar):
        if op == "=":
            n_match += n
            block += n
        elif op in "XID":
            block += n
    return n_match, block


def format_paf_line(qname, qlen, qstart, qend, strand, tname, tlen, tstart, tend, score, cigar,
                    primary=True, mapq=60):
    """One PAF line as minimap2 -c --eqx writes it.  qstart/qend are on the ALIGNED strand of the
    query (what the DP reports); for '-' records they are converted back to the forward read,
    which is what the PAF holds (and what PAF.__init__ flips again)."""
    if strand == "-":
        qstart, qend = qlen - qend, qlen - qstart
    n_match, block = cigar_counts(cigar)
    return "\t".join(map(str, [qname, qlen, qstart, qend, strand, tname, tlen, tstart, tend, n_match, block, mapq,
                               "tp:A:P" if primary else "tp:A:S", f"AS:i:{score}", f"cg:Z:{cigar}"]))

"""PAF records: the wire format between the reference's aligner and its selectors
(paf.py:32-79), kept for compatibility artefacts -- the hot path itself
exchanges integers, not text."""
import re


# the 12 mandatory PAF columns: (attribute, converter), in file order
_COLUMNS = (("qname", str), ("qlen", int), ("qstart", int), ("qend", int), ("strand", str), ("tname", str),
            ("tlen", int), ("tstart", int), ("tend", int), ("n_match", int), ("align_len", int), ("mapq", int))
# optional tags the selectors read: tag prefix -> (attribute, converter)
_TAGS = {"AS:i:": ("align_score", int), "cg:Z:": ("cigar", str), "tp:A:": ("is_primary", lambda v: v == "P")}


class PAF:
    """One PAF record with the attribute names the reference's selectors use (paf.py:32-79): the 12 fixed
    columns, `align_score` (AS:i, default -1), `cigar` (cg:Z), `is_primary` (tp:A:P).  On '-' records
    qstart/qend are turned onto the read's own strand (paf.py:70-74)."""

    def __init__(self, col_list):
        if len(col_list) < len(_COLUMNS):
            raise ValueError("number of columns should be >= 12 in a PAF file: " + "\t".join(col_list))
        for (name, conv), text in zip(_COLUMNS, col_list):
            setattr(self, name, conv(text))
        self.align_score, self.cigar, self.is_primary, self.tscore = -1, "", False, 0
        for tag in col_list[len(_COLUMNS):]:
            known = _TAGS.get(tag[:5])
            if known and (known[0] != "is_primary" or tag[5:] in ("P", "S")):
                setattr(self, known[0], known[1](tag[5:]))
        if self.strand not in ("+", "-"):
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

"""Rows upstream of the hot path: anchor finding, core trimming and the
round-1 / round-2 estimates that produce the path's inputs (core sequence, k window).

Mirrors nanoRepeat_bam.py:165-393 with the reference's names.  The two aligner calls --
anchors vs region reads (`pymm2.main` at :281) and cores vs `left + unit*T` (:362) -- become
one `nra_align_pairs` call each (the same DP engine as the path).  Differences from a
minimap2 run, all consequences of scoring an optimal DP instead of a heuristic mapper:
at most one record per (read, anchor, strand) -- no secondary hits -- and no mapping
quality, so `check_anchor_mapping`'s `mapq > 30` is taken as satisfied (mapq = 60).
"""
import numpy as np

from . import _capi
from .round3 import Read

_COMP = str.maketrans("ACGTNacgtn", "TGCANtgcan")


def rev_comp(seq):
    """tk.rev_comp (tk.py:346-357); N maps to N here (the reference raises KeyError, D3)."""
    return seq[::-1].translate(_COMP)


class AnchorHit:
    """The PAF fields find_anchor_locations_for1read reads (paf.py:32-79), with qstart/qend
    already on the read's own strand for '-' records, as PAF.__init__ leaves them (:70-74)."""

    def __init__(self, qname, qlen, qstart, qend, strand, tname, align_score, align_len, mapq=60):
        self.qname, self.qlen, self.qstart, self.qend = qname, qlen, qstart, qend
        self.strand, self.tname = strand, tname
        self.align_score, self.align_len, self.mapq = align_score, align_len, mapq


def check_anchor_mapping(hits):
    """Is the best hit of one anchor on one read unambiguous?  `hits` by decreasing score.  The rule of
    nanoRepeat_bam.py:165-179 as one predicate: a lone hit is accepted; otherwise the best must span at
    least 10 columns, beat the runner-up's score 1.5-fold and have mapq > 30."""
    if len(hits) < 2:
        return len(hits) == 1
    best, second = hits[0], hits[1]
    return best.align_len >= 10 and best.align_score > 1.5 * second.align_score and best.mapq > 30


CORE_BUFFER = 100       # bases of flank kept on either side of the core (nanoRepeat_bam.py:221)
MAX_READ_LEN = 4000000  # a read is the DP target of the anchors: NRA_MAX_TLEN_WIDE columns
MAX_CORE_LEN = 200000   # a core is the DP query of the round-2 template: NRA_MAX_QLEN rows


def _within_limit(repeat_region, sequences, limit, what):
    """The entries of {name: sequence} the C ABI takes (length <= limit).  One over-long read must not
    fail the call for every read of every region (the reference has no such limit): it is left out,
    stays without an estimate, and is recorded on the region for the pipeline's report."""
    keep = {}
    for name, seq in sequences.items():
        if len(seq.strip()) > limit:
            skipped = getattr(repeat_region, "skipped_reads", None)
            if skipped is None:
                skipped = repeat_region.skipped_reads = {}
            skipped[name] = f"{what} of {len(seq.strip())} bases (limit {limit})"
        else:
            keep[name] = seq
    return keep


def find_anchor_locations_for1read(read_paf_list, repeat_region):
    """Places one read (nanoRepeat_bam.py:181-236): both anchors must map unambiguously and in order
    (gap > -10 when on one strand); then the read enters `repeat_region.read_dict` with its core
    (anchors' inner edges +- 100 bases) and middle coordinates, all on the read's own strand."""
    if not read_paf_list:
        return
    ranked = {side: sorted((p for p in read_paf_list if p.tname.startswith(side + "_anchor")),
                           key=lambda p: -p.align_score) for side in ("left", "right")}
    if not (check_anchor_mapping(ranked["left"]) and check_anchor_mapping(ranked["right"])):
        return
    left, right = ranked["left"][0], ranked["right"][0]
    # anchors on opposite strands keep a gap of 0 and pass, as upstream (:207-212)
    gap = right.qstart - left.qend if left.strand == right.strand else 0
    if gap <= -10:
        return
    read = Read()
    read.read_name, read.full_read_len = read_paf_list[0].qname, read_paf_list[0].qlen
    read.left_anchor_is_good = read.right_anchor_is_good = read.both_anchors_are_good = True
    read.left_anchor_paf, read.right_anchor_paf, read.dist_between_anchors = left, right, gap
    read.strand = "+" if left.strand == "+" else "-"
    read.mid_seq_start_pos, read.mid_seq_end_pos = left.qend, right.qstart
    read.core_seq_start_pos = max(0, left.qend - CORE_BUFFER)
    read.core_seq_end_pos = min(read.full_read_len, right.qstart + CORE_BUFFER)
    read.left_buffer_len = left.qend - read.core_seq_start_pos
    read.right_buffer_len = read.core_seq_end_pos - right.qstart
    repeat_region.buffer_len = CORE_BUFFER
    repeat_region.read_dict[read.read_name] = read


def find_anchor_locations_in_reads(data_type, repeat_region, num_cpu=1, region_reads=None, device=0,
                                   scoring=None, aligner=None):
    """nanoRepeat_bam.py:260-286.  `region_reads` = {read_name: sequence} in file order (the
    reference reads them from region_fq_file)."""
    aligner = aligner or _capi.align_pairs
    region_reads = region_reads if region_reads is not None else repeat_region.region_reads
    region_reads = _within_limit(repeat_region, region_reads, MAX_READ_LEN, "read")
    names = list(region_reads)
    if not names:
        return
    # sequences: 0 = left anchor, 1 = right anchor, then read i forward at 2+2i, reverse at 3+2i.
    # The anchor is the DP's query (<= 3072 rows), the read its target (columns), so the extents
    # come back in read coordinates -- what the reference takes from the PAF's query columns.
    seqs = [repeat_region.left_anchor_seq, repeat_region.right_anchor_seq]
    pq, pt = [], []
    for i, n in enumerate(names):
        s = region_reads[n].strip()
        seqs += [s, rev_comp(s)]
        for a in (0, 1):
            for o in (0, 1):
                pq.append(a); pt.append(2 + 2 * i + o)
    out = aligner(seqs, np.array(pq, np.int32), np.array(pt, np.int32), sc=scoring, device=device)
    for i, n in enumerate(names):
        recs = []
        qlen = len(region_reads[n].strip())
        for a, tname in ((0, "left_anchor"), (1, "right_anchor")):
            for o, strand in ((0, "+"), (1, "-")):
                j = 4 * i + 2 * a + o
                if out["score"][j] < 0:
                    continue
                ts, te = int(out["tstart"][j]), int(out["tend"][j])
                recs.append(AnchorHit(n, qlen, ts, te, strand, tname, int(out["score"][j]), te - ts))
        find_anchor_locations_for1read(recs, repeat_region)


def make_core_seq(repeat_region, region_reads=None):
    """nanoRepeat_bam.py:288-331 without the FASTQ files: fills read_core_seq_dict."""
    region_reads = region_reads if region_reads is not None else repeat_region.region_reads
    for read_name, seq in region_reads.items():
        if read_name not in repeat_region.read_dict:
            continue
        read = repeat_region.read_dict[read_name]
        read_sequence = seq.strip()
        if read.strand == "-":
            read_sequence = rev_comp(read_sequence)
        repeat_region.read_core_seq_dict[read_name] = read_sequence[read.core_seq_start_pos:read.core_seq_end_pos]


def round1_and_round2_estimation(data_type, repeat_region, num_cpu=1, device=0, scoring=None, aligner=None):
    """nanoRepeat_bam.py:334-393: round 1 = anchor distance / unit length; round 2 = where the core's
    alignment against left + unit*T ends."""
    if len(repeat_region.read_dict) == 0:
        return
    aligner = aligner or _capi.align_pairs
    unit = repeat_region.repeat_unit_seq
    round1 = []
    for read in repeat_region.read_dict.values():
        read.round1_repeat_size = float(read.dist_between_anchors) / len(unit)
        round1.append(read.round1_repeat_size)
    template_repeat_size = int(max(round1) * 1.5) + 1
    if template_repeat_size < max(round1) + 10:
        template_repeat_size = int(max(round1) + 10)
    left = repeat_region.left_anchor_seq
    cores = _within_limit(repeat_region, repeat_region.read_core_seq_dict, MAX_CORE_LEN, "core")
    names = [n for n in repeat_region.read_dict if n in cores]
    seqs = [left + unit * template_repeat_size] + [repeat_region.read_core_seq_dict[n] for n in names]
    out = aligner(seqs, np.arange(1, len(seqs), dtype=np.int32), np.zeros(len(names), np.int32),
                  sc=scoring, device=device)
    for i, n in enumerate(names):
        if out["score"][i] < 0:
            continue
        ts, te = int(out["tstart"][i]), int(out["tend"][i])
        if ts <= len(left) and te >= len(left):                              # :371
            repeat_region.read_dict[n].round2_repeat_size = float(te - len(left)) / len(unit)


# ---------------------------------------------------------------------------------------------
# many regions per aligner call (a call costs a few milliseconds however small it is)
# ---------------------------------------------------------------------------------------------
MAX_BASES_PER_CALL = 1 << 28


def _chunks_by_bases(sizes, limit):
    """Consecutive index ranges whose sizes add up to at most `limit` (at least one item each)."""
    out, lo, acc = [], 0, 0
    for i, sz in enumerate(sizes):
        if i > lo and acc + sz > limit:
            out.append((lo, i)); lo, acc = i, 0
        acc += sz
    if lo < len(sizes):
        out.append((lo, len(sizes)))
    return out


def find_anchor_locations_in_reads_many(data_type, repeat_regions, reads_by_region, num_cpu=1, device=0,
                                        scoring=None, aligner=None, max_bases=MAX_BASES_PER_CALL):
    """find_anchor_locations_in_reads for many regions with ONE aligner call per ~256 M bases:
    same pairs, same per-read rule, one set of device buffers instead of one per region."""
    aligner = aligner or _capi.align_pairs
    reads_by_region = [_within_limit(region, reads, MAX_READ_LEN, "read")
                       for region, reads in zip(repeat_regions, reads_by_region)]
    sizes = [2 * sum(len(s) for s in reads.values()) for reads in reads_by_region]
    for lo, hi in _chunks_by_bases(sizes, max_bases):
        seqs, pq, pt, where = [], [], [], []
        for g in range(lo, hi):
            region, reads = repeat_regions[g], reads_by_region[g]
            base = len(seqs)
            seqs += [region.left_anchor_seq, region.right_anchor_seq]
            where.append((g, len(pq), list(reads)))
            for i, n in enumerate(reads):
                s = reads[n].strip()
                seqs += [s, rev_comp(s)]
                for a in (0, 1):
                    for o in (0, 1):
                        pq.append(base + a); pt.append(base + 2 + 2 * i + o)
        if not pq:
            continue
        out = aligner(seqs, np.array(pq, np.int32), np.array(pt, np.int32), sc=scoring, device=device)
        for g, p0, names in where:
            region, reads = repeat_regions[g], reads_by_region[g]
            for i, n in enumerate(names):
                recs = []
                qlen = len(reads[n].strip())
                for a, tname in ((0, "left_anchor"), (1, "right_anchor")):
                    for o, strand in ((0, "+"), (1, "-")):
                        j = p0 + 4 * i + 2 * a + o
                        if out["score"][j] < 0:
                            continue
                        ts, te = int(out["tstart"][j]), int(out["tend"][j])
                        recs.append(AnchorHit(n, qlen, ts, te, strand, tname, int(out["score"][j]), te - ts))
                find_anchor_locations_for1read(recs, region)


def round1_and_round2_estimation_many(data_type, repeat_regions, num_cpu=1, device=0, scoring=None, aligner=None,
                                      max_bases=MAX_BASES_PER_CALL):
    """round1_and_round2_estimation for many regions with one aligner call per ~256 M bases."""
    aligner = aligner or _capi.align_pairs
    plans = []
    for region in repeat_regions:
        if len(region.read_dict) == 0:
            plans.append(None)
            continue
        unit = region.repeat_unit_seq
        round1 = []
        for read in region.read_dict.values():
            read.round1_repeat_size = float(read.dist_between_anchors) / len(unit)
            round1.append(read.round1_repeat_size)
        template_repeat_size = int(max(round1) * 1.5) + 1
        if template_repeat_size < max(round1) + 10:
            template_repeat_size = int(max(round1) + 10)
        cores = _within_limit(region, region.read_core_seq_dict, MAX_CORE_LEN, "core")
        names = [n for n in region.read_dict if n in cores]
        plans.append((region.left_anchor_seq + unit * template_repeat_size, names))
    sizes = [0 if p is None else len(p[0]) + sum(len(repeat_regions[g].read_core_seq_dict[n]) for n in p[1])
             for g, p in enumerate(plans)]
    for lo, hi in _chunks_by_bases(sizes, max_bases):
        seqs, pq, pt, where = [], [], [], []
        for g in range(lo, hi):
            if plans[g] is None:
                continue
            template, names = plans[g]
            base = len(seqs)
            seqs.append(template)
            seqs += [repeat_regions[g].read_core_seq_dict[n] for n in names]
            where.append((g, len(pq), names))
            pq += list(range(base + 1, base + 1 + len(names))); pt += [base] * len(names)
        if not pq:
            continue
        out = aligner(seqs, np.array(pq, np.int32), np.array(pt, np.int32), sc=scoring, device=device)
        for g, p0, names in where:
            region = repeat_regions[g]
            left_len, unit_len = len(region.left_anchor_seq), len(region.repeat_unit_seq)
            for i, n in enumerate(names):
                j = p0 + i
                if out["score"][j] < 0:
                    continue
                ts, te = int(out["tstart"][j]), int(out["tend"][j])
                if ts <= left_len and te >= left_len:                            # :371
                    region.read_dict[n].round2_repeat_size = float(te - left_len) / unit_len

"""2D (joint) host side: mirrors the reference's round-2 / round-3 grid interface
(nanoRepeat_joint.py:234-507).

The functions keep the reference's names, arguments and results; the aligner loop -- one
`pymm2.main(...)` call, one template file and one FASTQ per grid cell -- and the Python
CIGAR walk of `estimate_two_repeats_from_paf` are replaced by ONE call through the C ABI
(`nra_joint_2d`) per round: the (read, k1, k2) list goes in, per-read tie sums come back.
"""
import numpy as np

from . import _capi


class Repeat:
    """nanoRepeat_joint.py:42-69."""

    def __init__(self):
        self.repeat_id = ""
        self.chrom = ""
        self.start = -1
        self.end = -1
        self.repeat_unit = ""
        self.repeat_unit_size = 0
        self.min_size = 0
        self.max_size = 1000
        self.round1_min_size = 0
        self.round1_max_size = 1000

    def init_from_string(self, string):
        col_list = string.split(":")
        if len(col_list) != 5:
            raise ValueError("--repeat1 and --repeat2 should be in this format: "
                             "chr:start:end:repeat_unit:max_size")
        self.chrom, self.start, self.end, self.repeat_unit, self.max_size = col_list
        self.start = int(self.start)
        self.end = int(self.end)
        self.repeat_unit_size = len(self.repeat_unit)
        self.min_size = 0
        self.max_size = int(self.max_size)
        self.repeat_id = "-".join(col_list[0:4])
        return self


class Round1Estimation:
    """nanoRepeat_joint.py:71-84: per-read [min, max) ranges of both repeat counts."""

    def __init__(self):
        self.repeat1_count_range_dict = dict()
        self.repeat2_count_range_dict = dict()
        self.potential_repeat_region_dict = dict()
        self.bad_reads_set = set()
        self.read_strand_dict = dict()      # not in the reference: +1 / -1 from round 1, saves the strand probe


class RepeatSize:
    """nanoRepeat_joint.py:86-91."""

    def __init__(self):
        self.repeat1_count_dict = dict()
        self.repeat2_count_dict = dict()
        self.step_size1 = 1
        self.step_size2 = 1


def choose_best_step_size(repeat, count_range_dict):
    """nanoRepeat_joint.py:351-374 (first minimum wins: the sort is stable)."""
    max_len = 50
    max_step_size = int(max_len / repeat.repeat_unit_size)
    if max_step_size < 1:
        max_step_size = 1
    error_list = [b - a for a, b in count_range_dict.values()]
    mean_range = np.mean(error_list)
    best_size, best_count = None, None
    for size in range(1, max_step_size + 1):
        count = int(mean_range / size) + 1
        count += size * 2 + 2
        if best_count is None or count < best_count:
            best_size, best_count = size, count
    return best_size


def extract_anchor_seq_for_two_repeats(repeat_chrom_seq, repeat1, repeat2, max_flanking_len):
    """nanoRepeat_joint.py:480-497."""
    left_end_pos = repeat1.start
    left_start_pos = max(0, left_end_pos - max_flanking_len)
    right_start_pos = repeat2.end
    right_end_pos = min(len(repeat_chrom_seq), right_start_pos + max_flanking_len)
    return (repeat_chrom_seq[left_start_pos:left_end_pos],
            repeat_chrom_seq[repeat1.end:repeat2.start],
            repeat_chrom_seq[right_start_pos:right_end_pos])


def build_template_for_two_repeats(left_anchor_seq, mid_anchor_seq, right_anchor_seq, repeat1,
                                   repeat2, repeat_count1, repeat_count2):
    """Name and sequence of one grid-cell template (nanoRepeat_joint.py:499-507)."""
    return ("%d-%d" % (repeat_count1, repeat_count2),
            left_anchor_seq + repeat1.repeat_unit * repeat_count1 + mid_anchor_seq +
            repeat2.repeat_unit * repeat_count2 + right_anchor_seq)


def calculate_repeat_size_from_exact_match(cigar, tstart, ref_repeat_start_pos, repeat_unit_size):
    """tk.py:405-431: whole units inside every run of exact matches that lies in (or reaches
    into) the repeat part of a template -- a deliberately low estimate of the repeat count."""
    from .paf import cigar_ops
    repeat_size = 0
    pos = tstart
    for op, n in cigar_ops(cigar):
        if op == "=":
            inside = n if pos >= ref_repeat_start_pos else pos + n - ref_repeat_start_pos
            if inside > 0:
                repeat_size += inside // repeat_unit_size
            pos += n
        elif op in "XD":
            pos += n
        elif op != "I":
            raise ValueError(f"unsupported CIGAR operation: {op}")
    return repeat_size


def round1_estimation_for1read(read_paf_list, repeat1, repeat2, left_anchor_len, right_anchor_len,
                               initial_estimation):
    """nanoRepeat_joint.py:591-649.  A read is used when exactly one primary, mapq >= 30 record
    per template spans that template's anchor/repeat boundary and the two are on opposite
    strands (the right template is reverse-complemented).  Upper bound of a count: how far the
    alignment reaches into the repeat, + 5; lower bound: min(max(0, e - 20), e // 2) with e the
    exact-match estimate."""
    spans = {"left_": [], "right": []}
    bounds = {"left_": left_anchor_len, "right": right_anchor_len}
    for paf in read_paf_list:
        if paf.mapq < 30 or not paf.is_primary:
            continue
        key = paf.tname[0:5]
        if key not in spans:
            raise ValueError(f"unknown template: {paf.tname}")
        if paf.tstart <= bounds[key] <= paf.tend:
            spans[key].append(paf)
    if len(spans["left_"]) != 1 or len(spans["right"]) != 1:
        return
    left_paf, right_paf = spans["left_"][0], spans["right"][0]
    if left_paf.strand == right_paf.strand:
        return
    readname = left_paf.qname
    for paf, bound, repeat, ranges in ((left_paf, left_anchor_len, repeat1, initial_estimation.repeat1_count_range_dict),
                                       (right_paf, right_anchor_len, repeat2, initial_estimation.repeat2_count_range_dict)):
        upper = int((paf.tend - bound) / repeat.repeat_unit_size) + 5
        exact = calculate_repeat_size_from_exact_match(paf.cigar, paf.tstart, bound, repeat.repeat_unit_size)
        ranges[readname] = (min(max(0, exact - 20), int(exact / 2.0)), upper)
    if left_paf.strand == "+":
        candidate = (left_paf.qstart, right_paf.qlen - right_paf.qstart)
    else:
        candidate = (right_paf.qstart, left_paf.qlen - left_paf.qstart)
    if candidate[1] - candidate[0] <= 0:
        initial_estimation.bad_reads_set.add(readname)
    initial_estimation.potential_repeat_region_dict[readname] = candidate
    initial_estimation.read_strand_dict[readname] = 1 if left_paf.strand == "+" else -1


def round1_estimation_from_paf(paf_records, repeat1, repeat2, left_anchor_len, right_anchor_len):
    """nanoRepeat_joint.py:561-589 on parsed records grouped by read name (the reference sorts the
    PAF text by its first column and groups consecutive lines)."""
    initial_estimation = Round1Estimation()
    by_read = dict()
    for paf in paf_records:
        by_read.setdefault(paf.qname, []).append(paf)
    for readname in sorted(by_read):
        round1_estimation_for1read(by_read[readname], repeat1, repeat2, left_anchor_len, right_anchor_len,
                                   initial_estimation)
    for readname in initial_estimation.bad_reads_set:
        initial_estimation.repeat1_count_range_dict.pop(readname, None)
        initial_estimation.repeat2_count_range_dict.pop(readname, None)
    return initial_estimation


def round1_templates(repeat_chrom_seq, repeat1, repeat2, max_anchor_len):
    """Names and sequences of the two round-1 templates (nanoRepeat_joint.py:515-538): the left
    anchor followed by max_size units of repeat 1, and the reverse complement of max_size units
    of repeat 2 followed by the right anchor."""
    from .upstream import rev_comp
    left_anchor_seq = repeat_chrom_seq[max(0, repeat1.start - max_anchor_len):repeat1.start]
    right_anchor_seq = repeat_chrom_seq[repeat2.end:min(len(repeat_chrom_seq), repeat2.end + max_anchor_len)]
    left = ("left_anchor_%d_%d_%s" % (len(left_anchor_seq), repeat1.max_size, repeat1.repeat_unit),
            left_anchor_seq + repeat1.repeat_unit * repeat1.max_size)
    right = ("right_anchor_%d_%d_%s_revc" % (len(right_anchor_seq), repeat2.max_size, repeat2.repeat_unit),
             rev_comp(repeat2.repeat_unit * repeat2.max_size + right_anchor_seq))
    return left, right, len(left_anchor_seq), len(right_anchor_seq)


def initial_estimate_repeat_size(repeat_chrom_seq, fastq_dict, data_type, num_threads, repeat1, repeat2,
                                 max_anchor_len, out_dir=None, device=0, scoring=None, aligner=None,
                                 cigar_aligner=None, save_paf=False):
    """Drop-in for nanoRepeat_joint.initial_estimate_repeat_size (nanoRepeat_joint.py:509-559); takes
    the reads as the dict `fastq_file_to_dict` returns instead of a file name.  The two aligner
    calls become: one score-only `nra_align_pairs` call that picks each read's strand per
    template (the lower strand would be a secondary record, which :602 drops), then one
    `nra_align_pairs_cigar` call for the chosen (read, template) pairs -- the CIGAR is needed for
    the exact-match estimate.  mapq is taken as 60."""
    from .paf import PAF, format_paf_line
    from .upstream import rev_comp
    assert repeat1.chrom == repeat2.chrom and repeat1.start < repeat2.start
    aligner = aligner or _capi.align_pairs
    cigar_aligner = cigar_aligner or _capi.align_pairs_cigar_chunked
    left, right, left_anchor_len, right_anchor_len = round1_templates(repeat_chrom_seq, repeat1, repeat2, max_anchor_len)
    names = list(fastq_dict)
    seqs = [left[1], right[1]]
    swap_ops = str.maketrans("ID", "DI")
    # A read is the DP's query (rows) and the template its target, as in a PAF record -- unless the
    # read is longer than the kernels hold in registers: then the template is the query and the
    # read the target (same score; extents and CIGAR are swapped back below; only the choice among
    # co-optimal paths can differ from the read-as-query alignment).
    long_read = []
    pq, pt = [], []
    for i, n in enumerate(names):
        s = _read_seq(fastq_dict[n])
        seqs += [s, rev_comp(s)]
        long_read.append(len(s) > MAX_READ_2D)
        for t in (0, 1):
            for o in (0, 1):
                a, b = 2 + 2 * i + o, t
                if long_read[i]:
                    a, b = b, a
                pq.append(a); pt.append(b)
    records = []
    if names:
        probe = aligner(seqs, np.array(pq, np.int32), np.array(pt, np.int32), sc=scoring, device=device)
        cq, ct, who = [], [], []
        for i, n in enumerate(names):
            for t in (0, 1):
                fwd, rev = int(probe["score"][4 * i + 2 * t]), int(probe["score"][4 * i + 2 * t + 1])
                if max(fwd, rev) < 0:
                    continue
                o = 0 if fwd >= rev else 1
                a, b = 2 + 2 * i + o, t
                if long_read[i]:
                    a, b = b, a
                cq.append(a); ct.append(b); who.append((n, i, t, "+-"[o], 2 + 2 * i + o))
        got = cigar_aligner(seqs, cq, ct, sc=scoring, device=device) if cq else None
        for j, (n, i, t, strand, read_idx) in enumerate(who):
            if int(got["score"][j]) < 0:
                continue
            tname, tseq = (left, right)[t]
            qs, qe, ts, te = (int(got[k][j]) for k in ("qstart", "qend", "tstart", "tend"))
            cigar = got["cigar"][j]
            if long_read[i]:
                qs, qe, ts, te = ts, te, qs, qe
                cigar = cigar.translate(swap_ops)
            records.append(format_paf_line(n, len(seqs[read_idx]), qs, qe, strand, tname, len(tseq), ts, te,
                                           int(got["score"][j]), cigar))
    if save_paf and out_dir:
        with open(f"{out_dir}/round1.paf", "w") as f:
            f.write("".join(r + "\n" for r in sorted(records, key=lambda r: r.split("\t")[0])))
    pafs = [PAF(r.split("\t")) for r in records]
    return round1_estimation_from_paf(pafs, repeat1, repeat2, left_anchor_len, right_anchor_len)


def _read_seq(fastq_record):
    return fastq_record.split("\n")[1].strip()


MAX_READ_2D = 3072          # rows one wave holds (NRA_MAX_QLEN_1BLOCK): the limit of nra_align_pairs_cigar queries.
                            # The grid rounds take reads of any length: nra_joint_2d scores longer ones cell by
                            # cell in chained row blocks, the full read like the reference (nanoRepeat_joint.py:332,408)


def _grid_product(k1_values, k2_values):
    """All (k1, k2) pairs, k1-major: the order the reference's nested loops list a read's cells in."""
    return np.repeat(k1_values, len(k2_values)), np.tile(k2_values, len(k1_values))


def _score_cells(region, readnames, fastq_dict, cells_by_read, device, scoring, scorer, strands, candidates=None):
    """One C-ABI call for every (read, k1, k2) cell; returns (RepeatSize, raw outputs)."""
    scorer = scorer or _capi.joint_2d
    names = [n for n in readnames if n in cells_by_read and len(cells_by_read[n][0])]
    reads = [_read_seq(fastq_dict[n]) for n in names]
    # cells of a read: (k1 array, k2 array), k1-major like the reference's nested grid loops
    per_read = [cells_by_read[n] for n in names]
    counts = [len(c[0]) for c in per_read]
    cell_read = np.repeat(np.arange(len(names), dtype=np.int32), counts)
    k1 = np.concatenate([c[0] for c in per_read]).astype(np.int32) if names else np.zeros(0, np.int32)
    k2 = np.concatenate([c[1] for c in per_read]).astype(np.int32) if names else np.zeros(0, np.int32)
    est = RepeatSize()
    if not names:
        return est, None
    st_in = None
    if strands is not None:
        st_in = np.array([strands.get(n, 0) for n in names], np.int8)
    out = scorer(region, reads, cell_read, k1, k2, read_strand=st_in, sc=scoring, device=device)
    for i, n in enumerate(names):
        if int(out["status"][i]) == _capi.READ_OK:        # nanoRepeat_joint.py:473-476
            nt = np.float64(out["n_ties"][i])
            est.repeat1_count_dict[n] = np.float64(out["sum_k1"][i]) / nt
            est.repeat2_count_dict[n] = np.float64(out["sum_k2"][i]) / nt
        if strands is not None:
            strands[n] = int(out["read_strand"][i])
    return est, out


def round2_estimation_of_repeat_size(initial_estimation, fastq_dict, repeat_chrom_seq, repeat1,
                                     repeat2, data_type="ont", num_threads=1, out_dir=None,
                                     device=0, scoring=None, scorer=None, strands=None):
    """Coarse grid (nanoRepeat_joint.py:376-425).  `strands` (dict, optional) carries each
    read's orientation between rounds so round 3 does not probe it again."""
    assert repeat1.chrom == repeat2.chrom
    assert repeat1.start < repeat2.start
    max_flanking_len = 1000
    step_size1 = choose_best_step_size(repeat1, initial_estimation.repeat1_count_range_dict)
    step_size2 = choose_best_step_size(repeat2, initial_estimation.repeat2_count_range_dict)
    left, mid, right = extract_anchor_seq_for_two_repeats(repeat_chrom_seq, repeat1, repeat2, max_flanking_len)
    # The reference loops grid cell by grid cell over all reads (:397-409); per read that is the
    # product of the grid values inside its two round-1 ranges, k1-major.
    grid1 = np.arange(repeat1.round1_min_size, repeat1.round1_max_size + 1, step_size1)
    grid2 = np.arange(repeat2.round1_min_size, repeat2.round1_max_size + 1, step_size2)
    cells = {}
    for readname in fastq_dict:
        if readname not in initial_estimation.repeat1_count_range_dict: continue
        if readname not in initial_estimation.repeat2_count_range_dict: continue
        min1, max1 = initial_estimation.repeat1_count_range_dict[readname]
        min2, max2 = initial_estimation.repeat2_count_range_dict[readname]
        cells[readname] = _grid_product(grid1[(grid1 >= min1) & (grid1 < max1)], grid2[(grid2 >= min2) & (grid2 < max2)])
    region = (left, repeat1.repeat_unit, mid, repeat2.repeat_unit, right)
    est, _ = _score_cells(region, list(fastq_dict), fastq_dict, cells, device, scoring, scorer, strands,
                          getattr(initial_estimation, "potential_repeat_region_dict", None))
    est.step_size1 = step_size1
    est.step_size2 = step_size2
    return est


def round3_estimation_of_repeat_size(initial_estimation, round2_estimation, fastq_dict,
                                     repeat_chrom_seq, repeat1, repeat2, data_type="ont",
                                     num_threads=1, out_dir=None, device=0, scoring=None,
                                     scorer=None, strands=None):
    """Fine grid, step 1 (nanoRepeat_joint.py:275-349)."""
    if len(round2_estimation.repeat1_count_dict) == 0 or len(round2_estimation.repeat2_count_dict) == 0:
        return RepeatSize()
    assert repeat1.chrom == repeat2.chrom
    assert repeat1.start < repeat2.start
    max_flanking_len = 1000
    buffer_size1 = round2_estimation.step_size1
    buffer_size2 = round2_estimation.step_size2
    size1_list, size2_list = [], []
    for readname in round2_estimation.repeat1_count_dict:
        if readname not in round2_estimation.repeat2_count_dict: continue
        size1_list.append(round2_estimation.repeat1_count_dict[readname])
        size2_list.append(round2_estimation.repeat2_count_dict[readname])
    min_size1 = max(0, int(min(size1_list) - buffer_size1))               # :298-303
    max_size1 = int(max(size1_list) + buffer_size1 + 2)
    min_size2 = max(0, int(min(size2_list) - buffer_size2))
    max_size2 = int(max(size2_list) + buffer_size2 + 2)
    left, mid, right = extract_anchor_seq_for_two_repeats(repeat_chrom_seq, repeat1, repeat2, max_flanking_len)
    grid1 = np.arange(min_size1, max_size1)
    grid2 = np.arange(min_size2, max_size2)
    cells = {}
    for readname in fastq_dict:                                           # :320-330, per read
        if readname not in round2_estimation.repeat1_count_dict: continue
        if readname not in round2_estimation.repeat2_count_dict: continue
        size1 = round2_estimation.repeat1_count_dict[readname]
        size2 = round2_estimation.repeat2_count_dict[readname]
        r1min1, r1max1 = initial_estimation.repeat1_count_range_dict[readname]
        r1min2, r1max2 = initial_estimation.repeat2_count_range_dict[readname]
        keep1 = (grid1 >= size1 - buffer_size1) & (grid1 < size1 + buffer_size1) & (grid1 >= r1min1) & (grid1 < r1max1)
        keep2 = (grid2 >= size2 - buffer_size2) & (grid2 < size2 + buffer_size2) & (grid2 >= r1min2) & (grid2 < r1max2)
        cells[readname] = _grid_product(grid1[keep1], grid2[keep2])
    region = (left, repeat1.repeat_unit, mid, repeat2.repeat_unit, right)
    est, _ = _score_cells(region, list(fastq_dict), fastq_dict, cells, device, scoring, scorer, strands,
                          getattr(initial_estimation, "potential_repeat_region_dict", None))
    est.step_size1 = 1
    est.step_size2 = 1
    return est


def fine_tune_read_count(initial_estimation, fastq_dict, repeat_chrom_seq, repeat1, repeat2,
                         data_type="ont", num_threads=1, out_dir=None, device=0, scoring=None,
                         scorer=None):
    """nanoRepeat_joint.py:234-273.  Takes the FASTQ as the {readname: 4-line record} dict the
    reference builds at :264 (file ingestion is outside the hot path)."""
    assert repeat1.chrom == repeat2.chrom
    assert repeat1.start < repeat2.start
    repeat1.round1_max_size = 0
    repeat2.round1_max_size = 0
    repeat1.round1_min_size = repeat1.max_size
    repeat2.round1_min_size = repeat2.max_size
    for lo, hi in initial_estimation.repeat1_count_range_dict.values():
        repeat1.round1_max_size = max(repeat1.round1_max_size, hi)
        repeat1.round1_min_size = min(repeat1.round1_min_size, lo)
    for lo, hi in initial_estimation.repeat2_count_range_dict.values():
        repeat2.round1_max_size = max(repeat2.round1_max_size, hi)
        repeat2.round1_min_size = min(repeat2.round1_min_size, lo)
    repeat1.round1_max_size = min(repeat1.round1_max_size, repeat1.max_size)
    repeat2.round1_max_size = min(repeat2.round1_max_size, repeat2.max_size)
    # a read's strand is known from round 1 when that was run here (the left template is forward)
    strands = dict(getattr(initial_estimation, "read_strand_dict", {}))
    round2_estimation = round2_estimation_of_repeat_size(
        initial_estimation, fastq_dict, repeat_chrom_seq, repeat1, repeat2, data_type, num_threads,
        out_dir, device, scoring, scorer, strands)
    if round2_estimation.step_size1 > 1 and round2_estimation.step_size2 > 1:     # :268
        return round3_estimation_of_repeat_size(
            initial_estimation, round2_estimation, fastq_dict, repeat_chrom_seq, repeat1, repeat2,
            data_type, num_threads, out_dir, device, scoring, scorer, strands)
    return round2_estimation


def output_repeat_size_2d(in_fastq_file, repeat1_id, repeat2_id, out_prefix, repeat1_count_dict,
                          repeat2_count_dict):
    """`<out_prefix>.repeat_size.txt` of the joint mode (split_alleles.py:560-600).  Rows are
    sorted by the first size (stable); reads present on one axis only cannot come out of the
    selector (both dicts are filled together)."""
    rows = []
    seen = set()
    for readname in list(repeat1_count_dict) + list(repeat2_count_dict):
        if readname in seen:
            continue
        seen.add(readname)
        rows.append((readname, repeat1_count_dict[readname], repeat2_count_dict[readname]))
    rows.sort(key=lambda x: x[1])
    text = f"##Input_FASTQ={in_fastq_file}\n#Read_Name\t{repeat1_id}.Repeat_Size\t{repeat2_id}.Repeat_Size\n"
    text += "".join(f"{n}\t{a:.1f}\t{b:.1f}\n" for n, a, b in rows)
    if out_prefix:
        with open(f"{out_prefix}.repeat_size.txt", "w") as f:
            f.write(text)
    return {n: (a, b) for n, a, b in rows}, text

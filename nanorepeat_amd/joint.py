"""2D (joint) host side: mirrors the reference's round-2 / round-3 grid interface
(nanoRepeat_joint.py:234-507).

The functions keep the reference's names, arguments and results; the aligner loop -- one
`pymm2.main(...)` call, one template file and one FASTQ per grid cell -- and the Python
CIGAR walk of `estimate_two_repeats_from_paf` are replaced by ONE call through the C ABI
(`nra_joint_2d`) per round: the (read, k1, k2) list goes in, per-read tie sums come back.
"""
from dataclasses import dataclass, field

import numpy as np
from itertools import chain
from operator import itemgetter

from . import _capi


@dataclass
class Repeat:
    """One of the two repeats of a joint run: the record nanoRepeat_joint.py:42-69 keeps (same field
    names, so either object can be handed to the other code base)."""
    repeat_id: str = ""
    chrom: str = ""
    start: int = -1
    end: int = -1
    repeat_unit: str = ""
    repeat_unit_size: int = 0
    min_size: int = 0
    max_size: int = 1000
    round1_min_size: int = 0
    round1_max_size: int = 1000

    SPEC = "chr:start:end:repeat_unit:max_size"

    @classmethod
    def parse(cls, spec):
        """`--repeat1 chr4:3074876:3074933:CAG:200` -> Repeat."""
        parts = spec.split(":")
        if len(parts) != 5:
            raise ValueError(f"--repeat1 and --repeat2 should be in this format: {cls.SPEC}")
        chrom, start, end, unit, max_size = parts
        return cls(repeat_id="-".join(parts[:4]), chrom=chrom, start=int(start), end=int(end), repeat_unit=unit,
                   repeat_unit_size=len(unit), min_size=0, max_size=int(max_size))

    def init_from_string(self, string):
        """The reference's spelling of parse(): fills this object in place and returns it."""
        self.__dict__.update(Repeat.parse(string).__dict__)
        return self


@dataclass
class Round1Estimation:
    """What round 1 knows per read (nanoRepeat_joint.py:71-84): [min, max) ranges of both repeat counts,
    where on the read the repeats lie, which reads to drop -- and, not in the reference, the strand, which
    saves the grid rounds their strand probe."""
    repeat1_count_range_dict: dict = field(default_factory=dict)
    repeat2_count_range_dict: dict = field(default_factory=dict)
    potential_repeat_region_dict: dict = field(default_factory=dict)
    bad_reads_set: set = field(default_factory=set)
    read_strand_dict: dict = field(default_factory=dict)      # +1 / -1


@dataclass
class RepeatSize:
    """Per-read estimates of one grid round and the grid steps they were made with (nanoRepeat_joint.py:86-91)."""
    repeat1_count_dict: dict = field(default_factory=dict)
    repeat2_count_dict: dict = field(default_factory=dict)
    step_size1: int = 1
    step_size2: int = 1


def _all_ranges(count_range_dict):
    """Every [min, max) of a round-1 range dict as an (n, 2) int64 array, in the dict's order (one flat pass)."""
    return np.fromiter(chain.from_iterable(count_range_dict.values()), np.int64, 2 * len(count_range_dict)).reshape(-1, 2)


def choose_best_step_size(repeat, count_range_dict, _ranges=None):
    """nanoRepeat_joint.py:351-374 (first minimum wins: the sort is stable).  `_ranges`: the dict's values as an array,
    when the caller has them already."""
    max_len = 50
    max_step_size = int(max_len / repeat.repeat_unit_size)
    if max_step_size < 1:
        max_step_size = 1
    span = _all_ranges(count_range_dict) if _ranges is None else _ranges
    mean_range = np.mean(span[:, 1] - span[:, 0])
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
    """The sequence line of a 4-line FASTQ record (two index() calls instead of splitting all four lines)."""
    a = fastq_record.index("\n") + 1
    b = fastq_record.find("\n", a)
    return fastq_record[a:b if b >= 0 else None].strip()


MAX_READ_2D = 3072          # rows one wave holds (NRA_MAX_QLEN_1BLOCK): the limit of nra_align_pairs_cigar queries.
                            # The grid rounds take reads of any length: nra_joint_2d scores longer ones cell by
                            # cell in chained row blocks, the full read like the reference (nanoRepeat_joint.py:332,408)


PIPELINE_MIN_READS = 2000     # (round 2 of the build ran the reads as two groups from here on; see GridSession.parts)


class LazyRepeatSize(RepeatSize):
    """A RepeatSize whose two dicts are built from the per-read arrays of a grid round the first time someone
    looks at them.  Round 3 takes round 2's sizes as arrays (`sizes_for`), so when round 3 follows, round 2's
    10 000 dict entries are never made."""

    def __init__(self, names=None, size1=None, size2=None, step_size1=1, step_size2=1):
        object.__setattr__(self, "_names", names if names is not None else [])
        object.__setattr__(self, "_size1", size1)
        object.__setattr__(self, "_size2", size2)
        object.__setattr__(self, "_dicts", None)
        self.step_size1, self.step_size2 = step_size1, step_size2

    def _build(self):
        if self._dicts is None:
            if self._size1 is None:
                object.__setattr__(self, "_dicts", ({}, {}))
            else:       # np.float64 values, like np.mean's (nanoRepeat_joint.py:475-476)
                object.__setattr__(self, "_dicts", (dict(zip(self._names, self._size1)), dict(zip(self._names, self._size2))))
        return self._dicts

    repeat1_count_dict = property(lambda self: self._build()[0],
                                  lambda self, v: object.__setattr__(self, "_dicts", (v, self._build()[1])))
    repeat2_count_dict = property(lambda self: self._build()[1],
                                  lambda self, v: object.__setattr__(self, "_dicts", (self._build()[0], v)))

    def __len__(self):
        return len(self._names) if self._dicts is None else len(self._dicts[0])

    def sizes_for(self, names):
        """(rows, size1, size2): numbers (into `names`) of the reads with an estimate, and their sizes."""
        if self._dicts is None and self._size1 is not None and getattr(self, "_rows_of", None) is not None and \
                self._rows_of[0] is names:
            return self._rows_of[1], self._size1, self._size2
        return None


class GridSession:
    """The reads of a joint run for the grid rounds: packed and uploaded once (`nra_batch2d_create_reads`),
    scored against one routed grid per round (`nra_batch2d_set_grid` + run).  With an injected `scorer`
    (the tests' oracle twin of `_capi.joint_2d`) every round is a one-shot call on the grid's cell list instead.

    `parts` > 1 (optional; default 1): the reads are cut into that many contiguous groups, each a session of its
    own (`subs`) with its own resident batch and stream, and `fine_tune_read_count` runs the groups in host
    threads that take turns: a thread holds `host_lock` while it works on the host and lets go of it while it
    waits for the device.  That hid the host's share of a round when the cell lists were built in numpy (round 2
    of the build: 12 of 22 ms); with the routing in the library (`nra_batch2d_set_grid`) one group is faster."""

    def __init__(self, region, fastq_dict, device=0, scoring=None, scorer=None, parts=None, flags=0):
        self.region = region
        self.names = list(fastq_dict)
        self.device, self.scoring, self.scorer = device, scoring, scorer
        self.batch = None
        self.refined = False        # the last score_grid carried the refinement it was asked for
        self._rounds = None         # `rounds`: set to a list to collect (n_cells, batch statistics, group) of every round
        self.subs, self.pool, self.host_lock = [], None, None
        self.in_turn = False        # this group's thread holds host_lock (fine_tune_read_count)
        parts = max(1, min(int(parts or 1), max(1, len(self.names))))
        if parts > 1:
            from concurrent.futures import ThreadPoolExecutor
            cuts = [len(self.names) * p // parts for p in range(parts + 1)]
            self.subs = [GridSession(region, {n: fastq_dict[n] for n in self.names[a:b]}, device, scoring, scorer, parts=1, flags=flags)
                         for a, b in zip(cuts[:-1], cuts[1:])]
            self.pool = ThreadPoolExecutor(max_workers=parts, thread_name_prefix="nra-grid")
            import threading
            self.host_lock = threading.Lock()
            for sub in self.subs:
                sub.host_lock = self.host_lock
            return
        self.index = {n: i for i, n in enumerate(self.names)}
        self.reads = [_read_seq(fastq_dict[n]) for n in self.names]
        if scorer is None:
            self.batch = _capi.Batch.create_2d_reads(region, self.reads, sc=scoring, flags=flags, device=device)

    @property
    def rounds(self):
        return self._rounds

    @rounds.setter
    def rounds(self, value):        # the read groups append to the same list
        self._rounds = value
        for sub in self.subs:
            sub.rounds = value

    def score_grid(self, grid, read_strand, refine=None):
        """One grid round for this session's reads -> per-read tie sums (dict of arrays) and the number of cells.
        refine = (buf1, buf2, lo1, hi1, lo2, hi2): enqueue the reference's round 3 behind the grid's run
        (nra_batch2d_refine) -- the results are then the refinement's and `self.refined` says so; where the batch cannot
        (or with an oracle scorer), the grid's own results come back and the caller runs round 3 as a grid of its own."""
        self.refined = False
        if self.scorer is not None:
            cell_read, k1, k2 = _capi.joint_grid_cells(grid)
            if len(cell_read) == 0:
                return None, 0
            return self.scorer(self.region, self.reads, cell_read, k1, k2, read_strand=read_strand, sc=self.scoring,
                               device=self.device), len(cell_read)
        n_cells = self.batch.set_grid(grid, read_strand)
        if n_cells == 0:
            return None, 0
        self.batch.run()
        if refine is not None:
            self.refined = self.batch.refine(*refine)
        if self.in_turn:
            self.host_lock.release()               # another group's thread works on the host meanwhile
            try:
                self.batch.sync()
            finally:
                self.host_lock.acquire()
        else:
            self.batch.sync()
        if self.rounds is not None or self.refined:
            st = self.batch.stats()
            n_cells = int(st["n_alignments"])              # (both grids' cells when a refinement ran)
            if self.rounds is not None:
                self.rounds.append((n_cells, st, id(self)))
        return self.batch.fetch(per_candidate=False), n_cells

    def sweep_flanks(self, read_strand_dict):
        """The strand-only flank sweeps of the coming grid rounds, enqueued now (nra_batch2d_sweep_flanks): the device works
        while the host derives ranges, step sizes and grids.  Reads without a known strand are left to their cell list."""
        for sub in self.subs:
            sub.sweep_flanks(read_strand_dict)
        if self.batch is None or not read_strand_dict:
            return
        n = len(self.names)
        if len(read_strand_dict) == n and list(read_strand_dict) == self.names:
            strand = np.fromiter(read_strand_dict.values(), np.int8, n)
        else:
            strand = np.fromiter((read_strand_dict.get(name, 0) for name in self.names), np.int8, n)
        self.batch.sweep_flanks(strand)

    def new_run(self):
        """The grid rounds are about to start over on these reads: nothing of an earlier run is reused."""
        for sub in self.subs:
            sub.new_run()
        if self.batch is not None:
            self.batch.invalidate()

    def close(self):
        for sub in self.subs:
            sub.close()
        if self.pool is not None:
            self.pool.shutdown(wait=True)
            self.pool = None
        if self.batch is not None:
            self.batch.close()
            self.batch = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def _rows_with(names, *dicts):
    """Numbers of the names every dict holds (ascending).  One C-level membership pass per dict."""
    if all(len(d) >= len(names) and all(map(d.__contains__, names)) for d in dicts):
        return np.arange(len(names), dtype=np.int64)
    return np.array([i for i, name in enumerate(names) if all(name in d for d in dicts)], np.int64)


def _values_of(d, names, rows, dtype, width=1):
    """d[names[i]] for i in rows as an array of `width` columns.  When the dict lists exactly these names in this
    order (the usual case: both come from the same FASTQ) its values are read in one flat pass."""
    if len(rows) == 0:
        return np.zeros((0, width) if width > 1 else 0, dtype)
    if len(rows) == len(names) == len(d) and list(d) == names:
        flat = chain.from_iterable(d.values()) if width > 1 else d.values()
        out = np.fromiter(flat, dtype, len(names) * width)
    else:
        keys = names if len(rows) == len(names) else [names[i] for i in rows]
        got = itemgetter(*keys)(d)
        out = np.array(got if len(keys) > 1 else [got], dtype)
    return out.reshape(-1, width) if width > 1 else out


class _Round1Arrays:
    """The round-1 knowledge of a session's reads as arrays, made once per fine_tune_read_count call: `rows` = the
    session read numbers with a range on both axes, their [min, max) ranges, every read's strand (0 = unknown)."""

    def __init__(self, session, initial_estimation, strands, all_ranges=None):
        names = session.names
        r1, r2 = initial_estimation.repeat1_count_range_dict, initial_estimation.repeat2_count_range_dict
        self.source = initial_estimation
        # every read's ranges in dict order (spans and step sizes come from ALL reads, nanoRepeat_joint.py:239-259, :365)
        self.all1, self.all2 = all_ranges if all_ranges is not None else (_all_ranges(r1), _all_ranges(r2))
        self.rows = _rows_with(names, r1, r2)
        if len(self.rows) == len(names) == len(r1) == len(r2) and list(r1) == names and list(r2) == names:
            self.range1, self.range2 = self.all1, self.all2           # the session's reads in order: the same arrays
        else:
            self.range1 = _values_of(r1, names, self.rows, np.int64, 2)
            self.range2 = _values_of(r2, names, self.rows, np.int64, 2)
        self.strand = None
        if strands is not None:
            if len(strands) == len(names) and list(strands) == names:
                self.strand = np.fromiter(strands.values(), np.int8, len(names))
            else:
                self.strand = np.fromiter((strands.get(name, 0) for name in names), np.int8, len(names))


def _score_round(session, ctx, rows, axis1, lo1, hi1, axis2, lo2, hi2, strands, steps, refine=False):
    """One grid round: `rows` = session read numbers with bounds [lo, hi) on each axis.  Returns a RepeatSize.
    refine: also ask for the reference's round 3 behind it, on the device (buffers = this grid's steps, bounds = the same
    round-1 ranges); `session.refined` tells whether the result is the refinement's (steps 1, 1) or the grid's own."""
    if len(rows) == 0:
        return RepeatSize()
    n = len(session.names)
    full = [np.zeros(n, np.float64) for _ in range(4)]                 # reads without bounds: empty [0, 0)
    for dst, src in zip(full, (lo1, hi1, lo2, hi2)):
        dst[rows] = src
    out, n_cells = session.score_grid(_capi.Grid(axis1, full[0], full[1], axis2, full[2], full[3]), ctx.strand,
                                      (steps[0], steps[1], *full) if refine else None)
    if n_cells == 0:
        return RepeatSize()
    if refine and session.refined:
        steps = (1, 1)
    # a read with cells has status OK or NO_RECORD; the others are left at OK with no ties (nanoRepeat_joint.py:473-476)
    ok = np.nonzero((np.asarray(out["status"]) == _capi.READ_OK) & (np.asarray(out["n_ties"]) > 0))[0]
    nt = np.asarray(out["n_ties"], np.float64)[ok]
    names = session.names if len(ok) == n else [session.names[i] for i in ok]
    est = LazyRepeatSize(names, np.asarray(out["sum_k1"], np.float64)[ok] / nt, np.asarray(out["sum_k2"], np.float64)[ok] / nt,
                         *steps)
    object.__setattr__(est, "_rows_of", (session.names, ok))
    if strands is not None and (ctx.strand is None or (ctx.strand == 0).any()):
        got = np.asarray(out["read_strand"])
        idx = np.nonzero(got != 0)[0]
        strands.update(zip((session.names[i] for i in idx), got[idx].tolist()))
        ctx.strand = np.where(got != 0, got, ctx.strand if ctx.strand is not None else 0).astype(np.int8)
    return est


def _merge_parts(parts):
    """One RepeatSize from those of a session's read groups (in read order, like the un-split run's)."""
    est = RepeatSize()
    for e in parts:
        est.repeat1_count_dict.update(e.repeat1_count_dict)
        est.repeat2_count_dict.update(e.repeat2_count_dict)
    scored = next((e for e in parts if len(e.repeat1_count_dict) > 0), parts[0])
    est.step_size1, est.step_size2 = scored.step_size1, scored.step_size2
    return est


def _joint_region(repeat_chrom_seq, repeat1, repeat2, max_flanking_len=1000):
    left, mid, right = extract_anchor_seq_for_two_repeats(repeat_chrom_seq, repeat1, repeat2, max_flanking_len)
    return (left, repeat1.repeat_unit, mid, repeat2.repeat_unit, right)


def _check_repeat_order(repeat1, repeat2):
    if repeat1.chrom != repeat2.chrom or not repeat1.start < repeat2.start:
        raise AssertionError("the two repeats must lie on one chromosome, repeat1 first")


def _context(session, initial_estimation, strands, ctx):
    if ctx is not None and ctx.source is initial_estimation:
        return ctx
    return _Round1Arrays(session, initial_estimation, strands)


def _axis(lo, hi, step):
    """(start, step, count) of range(lo, hi, step)."""
    return (int(lo), int(step), max(0, -(-(int(hi) - int(lo)) // int(step))))


def round2_estimation_of_repeat_size(initial_estimation, fastq_dict, repeat_chrom_seq, repeat1,
                                     repeat2, data_type="ont", num_threads=1, out_dir=None,
                                     device=0, scoring=None, scorer=None, strands=None, session=None, _ctx=None,
                                     _refine=False):
    """Coarse grid (nanoRepeat_joint.py:376-425).  `strands` (dict, optional) carries each read's
    orientation between rounds so round 3 does not probe it again; `session` the resident reads.
    _refine (fine_tune_read_count): when both steps come out > 1 -- the reference then runs round 3 (:268) -- ask the
    library to run that round behind this one on the device (nra_batch2d_refine); if it does, the estimate returned is
    round 3's (steps 1, 1, `refined` set) and the caller skips its own round 3."""
    _check_repeat_order(repeat1, repeat2)
    if session is not None and session.subs:
        return _merge_parts(list(session.pool.map(lambda sub: round2_estimation_of_repeat_size(
            initial_estimation, fastq_dict, repeat_chrom_seq, repeat1, repeat2, data_type, num_threads, out_dir, device,
            scoring, scorer, strands, sub), session.subs)))
    have = _ctx if _ctx is not None and _ctx.source is initial_estimation else None
    step_size1 = choose_best_step_size(repeat1, initial_estimation.repeat1_count_range_dict, None if have is None else have.all1)
    step_size2 = choose_best_step_size(repeat2, initial_estimation.repeat2_count_range_dict, None if have is None else have.all2)
    own = session is None
    if own:
        session = GridSession(_joint_region(repeat_chrom_seq, repeat1, repeat2), fastq_dict, device, scoring, scorer, parts=1)
    try:
        # The reference visits grid cell by grid cell (k1 in range(min1, max1 + 1, s1), k2 alike, :397-398) and, per
        # cell, the reads whose round-1 ranges hold it (min <= k < max on both axes, :407): per read that is the
        # product of the grid values inside its two ranges -- the routing nra_batch2d_set_grid does
        ctx = _context(session, initial_estimation, strands, _ctx)
        est = _score_round(session, ctx, ctx.rows,
                           _axis(repeat1.round1_min_size, repeat1.round1_max_size + 1, step_size1), ctx.range1[:, 0], ctx.range1[:, 1],
                           _axis(repeat2.round1_min_size, repeat2.round1_max_size + 1, step_size2), ctx.range2[:, 0], ctx.range2[:, 1],
                           strands, (step_size1, step_size2), refine=_refine and step_size1 > 1 and step_size2 > 1)
        refined = bool(_refine and step_size1 > 1 and step_size2 > 1 and session.refined)
    finally:
        if own:
            session.close()
    if refined:
        est.step_size1 = est.step_size2 = 1                       # nanoRepeat_joint.py:345-346
        object.__setattr__(est, "refined", True)
    else:
        est.step_size1, est.step_size2 = step_size1, step_size2
    return est


def round3_estimation_of_repeat_size(initial_estimation, round2_estimation, fastq_dict,
                                     repeat_chrom_seq, repeat1, repeat2, data_type="ont",
                                     num_threads=1, out_dir=None, device=0, scoring=None,
                                     scorer=None, strands=None, session=None, _ctx=None):
    """Fine grid, step 1 (nanoRepeat_joint.py:275-349)."""
    if len(round2_estimation) == 0 if isinstance(round2_estimation, LazyRepeatSize) else \
            (len(round2_estimation.repeat1_count_dict) == 0 or len(round2_estimation.repeat2_count_dict) == 0):
        return RepeatSize()
    _check_repeat_order(repeat1, repeat2)
    if session is not None and session.subs:
        return _merge_parts(list(session.pool.map(lambda sub: round3_estimation_of_repeat_size(
            initial_estimation, round2_estimation, fastq_dict, repeat_chrom_seq, repeat1, repeat2, data_type, num_threads,
            out_dir, device, scoring, scorer, strands, sub), session.subs)))
    buf1, buf2 = round2_estimation.step_size1, round2_estimation.step_size2
    own = session is None
    if own:
        session = GridSession(_joint_region(repeat_chrom_seq, repeat1, repeat2), fastq_dict, device, scoring, scorer, parts=1)
    try:
        ctx = _context(session, initial_estimation, strands, _ctx)
        have = round2_estimation.sizes_for(session.names) if isinstance(round2_estimation, LazyRepeatSize) else None
        if have is not None:                     # round 2 was scored on this very session: its arrays, no dicts
            rows, size1, size2 = have
            all1, all2 = size1, size2
        else:
            done1, done2 = round2_estimation.repeat1_count_dict, round2_estimation.repeat2_count_dict
            rows = _rows_with(session.names, done1, done2)
            size1 = _values_of(done1, session.names, rows, np.float64)
            size2 = _values_of(done2, session.names, rows, np.float64)
            both = [name for name in done1 if name in done2]
            all1 = np.array([done1[name] for name in both], np.float64)
            all2 = np.array([done2[name] for name in both], np.float64)
        # the global grid spans every read's round-2 size +- one coarse step (:298-303) ...
        axis1 = _axis(max(0, int(all1.min() - buf1)), int(all1.max() + buf1 + 2), 1)
        axis2 = _axis(max(0, int(all2.min() - buf2)), int(all2.max() + buf2 + 2), 1)
        # ... and a read takes the cells within one step of its own size that lie inside its round-1 range (:320-330)
        where = np.zeros(0, np.int64)                      # the reads of `rows` all have round-1 ranges
        if len(rows):
            if len(ctx.rows) == 0:
                raise KeyError("a read with a round-2 size has no round-1 range")
            where = np.minimum(np.searchsorted(ctx.rows, rows), len(ctx.rows) - 1)
            if not np.array_equal(ctx.rows[where], rows):
                raise KeyError("a read with a round-2 size has no round-1 range")
        r1, r2 = ctx.range1[where], ctx.range2[where]
        est = _score_round(session, ctx, rows,
                           axis1, np.maximum(size1 - buf1, r1[:, 0]), np.minimum(size1 + buf1, r1[:, 1]),
                           axis2, np.maximum(size2 - buf2, r2[:, 0]), np.minimum(size2 + buf2, r2[:, 1]), strands, (1, 1))
    finally:
        if own:
            session.close()
    est.step_size1 = est.step_size2 = 1
    return est


def fine_tune_read_count(initial_estimation, fastq_dict, repeat_chrom_seq, repeat1, repeat2,
                         data_type="ont", num_threads=1, out_dir=None, device=0, scoring=None,
                         scorer=None, session=None, refine=True):
    """nanoRepeat_joint.py:234-273.  Takes the FASTQ as the {readname: 4-line record} dict the
    reference builds at :264 (file ingestion is outside the hot path).  The reads are packed and
    uploaded once for both grid rounds (`session`: a GridSession to reuse, e.g. a benchmark's).
    refine: let the library run round 3 behind round 2 on the device where it can (nra_batch2d_refine: one enqueue, one
    fetch, no host work between the rounds); False = two grid calls with the host routing round 3 -- same results."""
    _check_repeat_order(repeat1, repeat2)
    own = session is None
    if own:
        session = GridSession(_joint_region(repeat_chrom_seq, repeat1, repeat2), fastq_dict, device, scoring, scorer)
    try:
        # what the device sweeps first depends on the reads and their strands only: it goes out before any host work
        session.sweep_flanks(getattr(initial_estimation, "read_strand_dict", None))
        all_ranges = (_all_ranges(initial_estimation.repeat1_count_range_dict), _all_ranges(initial_estimation.repeat2_count_range_dict))
        for rep, span in ((repeat1, all_ranges[0]), (repeat2, all_ranges[1])):
            lo = min(rep.max_size, int(span[:, 0].min())) if len(span) else rep.max_size
            hi = max(0, int(span[:, 1].max())) if len(span) else 0
            rep.round1_min_size, rep.round1_max_size = lo, min(hi, rep.max_size)            # :239-259
    except BaseException:
        if own:
            session.close()
        raise

    def both_rounds(sess, rep1, rep2):
        # a read's strand is known from round 1 when that was run here (the left template is forward)
        strands = dict(getattr(initial_estimation, "read_strand_dict", {}))
        ctx = _Round1Arrays(sess, initial_estimation, strands, all_ranges)
        est = round2_estimation_of_repeat_size(initial_estimation, fastq_dict, repeat_chrom_seq, rep1, rep2,
                                               data_type, num_threads, out_dir, device, scoring, scorer, strands, sess, ctx,
                                               _refine=refine)
        if getattr(est, "refined", False):        # round 3 ran behind round 2 on the device: this is its estimate
            return est
        if est.step_size1 > 1 and est.step_size2 > 1:                                    # :268
            est = round3_estimation_of_repeat_size(initial_estimation, est, fastq_dict, repeat_chrom_seq, rep1,
                                                   rep2, data_type, num_threads, out_dir, device, scoring, scorer,
                                                   strands, sess, ctx)
        return est

    try:
        if not session.subs:
            return both_rounds(session, repeat1, repeat2)
        # Groups of reads in parallel threads (GridSession.parts).  Everything global -- the round-1 spans above, the
        # step sizes -- comes from the round-1 ranges of ALL reads, which every group sees; what a group derives
        # from its own round-2 results (round 3's grid span) only has to cover its own reads' cells.
        def turn(sub):
            with session.host_lock:
                sub.in_turn = True
                try:
                    return both_rounds(sub, repeat1, repeat2)
                finally:
                    sub.in_turn = False

        return _merge_parts(list(session.pool.map(turn, session.subs)))
    finally:
        if own:
            session.close()


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

"""1D host side: mirrors the reference's round-3 interface (nanoRepeat_bam.py:408-500).

`round3_estimation(data_type, fast_mode, repeat_region, num_cpu)` has the reference's
name, argument meaning and effect -- it sets `read.round3_repeat_size` on every read of
`repeat_region.read_dict` that has a round-2 estimate -- but the per-read
`pymm2.main(...)` loop (one aligner call and two temp files per read) is replaced by ONE
call through the C ABI (`nra_round3_1d`) for the whole region, or for many regions at once
with `round3_estimation_regions`.
"""
import numpy as np

from . import _capi

DATA_TYPES = ("ont", "ont_sup", "ont_q20", "clr", "hifi")   # all map to `-x map-ont` (tk.py:502-517)
MAX_CORE_LEN = 200000       # NRA_MAX_QLEN: cores above 3072 bases run as chained row blocks in int32 cells
MAX_TEMPLATE_LEN = 4000000  # NRA_MAX_TLEN_WIDE
READ_TOO_LONG = 4           # round3_status of a read left at its round-2 size because of those limits


class Read:
    """Fields of the reference's Read record that the path touches (repeat_region.py:32-56)."""

    def __init__(self, read_name=None, round2_repeat_size=None):
        self.read_name = read_name
        self.round1_repeat_size = None
        self.round2_repeat_size = round2_repeat_size
        self.round3_repeat_size = None
        self.round3_status = None       # NRA_READ_* of the last estimation (not in the reference)
        self.round3_best_score = None


class RepeatRegion:
    """Fields of the reference's RepeatRegion the path touches (repeat_region.py:116-193)."""

    def __init__(self, line=None, no_details=False):
        self.left_anchor_seq = None
        self.right_anchor_seq = None
        self.left_anchor_len = None
        self.right_anchor_len = None
        self.repeat_unit_seq = None
        self.chrom = None
        self.start_pos = None
        self.end_pos = None
        self.out_prefix = None
        self.no_details = no_details
        self.read_dict = dict()
        self.read_core_seq_dict = dict()
        self.region_fq_file = None      # reads of the region (source of the per-allele FASTQ files)
        self.results = None             # phasing.Result, filled by output_repeat_size_1d / step 4
        self.final_output = None
        self.index = None
        if line is not None:
            col_list = line.strip().split("\t")
            if len(col_list) < 4:
                raise ValueError("the repeat region bed file should be tab-delimited and have 4 columns: "
                                 "chrom, start_position, end_position, repeat_unit")
            self.chrom, self.start_pos, self.end_pos, self.repeat_unit_seq = col_list[0:4]
            self.start_pos = int(self.start_pos)
            self.end_pos = int(self.end_pos)

    def to_unique_id(self):
        return f"{self.chrom}-{self.start_pos}-{self.end_pos}-{self.repeat_unit_seq}"


def round3_window(round2_repeat_size, fast_mode):
    """Candidate window of one read: (kmin, kmax) inclusive.  nanoRepeat_bam.py:463-472."""
    buffer = max(15, int(round2_repeat_size * 0.05))
    if buffer > 150:
        buffer = 150
    if fast_mode:
        buffer = 15
    max_template_repeat_size = int(round2_repeat_size + buffer)
    min_template_repeat_size = int(round2_repeat_size - buffer)
    if min_template_repeat_size < 0:
        min_template_repeat_size = 0
    return min_template_repeat_size, max_template_repeat_size


def _check_data_type(data_type):
    if data_type not in DATA_TYPES:
        raise ValueError(f"Unknown data type: {data_type}")      # tk.py:514-516


def round3_estimation_regions(data_type, fast_mode, repeat_regions, num_cpu=1, device=0,
                              scoring=None, scorer=None):
    """Round 3 for a list of regions in one batch.  `scorer` is the C-ABI call
    (`_capi.round3_1d`); tests may inject a twin with the same signature."""
    _check_data_type(data_type)
    scorer = scorer or _capi.round3_1d
    regions, reads, kmin, kmax, rr, owners = [], [], [], [], [], []
    for g, region in enumerate(repeat_regions):
        regions.append((region.left_anchor_seq, region.repeat_unit_seq, region.right_anchor_seq))
        for read_name in region.read_dict:                       # nanoRepeat_bam.py:457
            read = region.read_dict[read_name]
            r2 = read.round2_repeat_size
            if r2 is None:                                       # :460 -- read skipped
                lo, hi = 0, -1
                seq = ""
            else:
                lo, hi = round3_window(r2, fast_mode)
                seq = region.read_core_seq_dict[read_name].strip()   # :487
                template_len = (len(region.left_anchor_seq) + len(region.repeat_unit_seq) * hi +
                                len(region.right_anchor_seq))
                if len(seq) > MAX_CORE_LEN or template_len > MAX_TEMPLATE_LEN:
                    # beyond what the C ABI takes (a core of > 200 kb): one such read must not fail the
                    # whole batch -- it keeps its round-2 size, like a read whose best records fail the
                    # flank test (nanoRepeat_bam.py:432-433), and is reported by the pipeline
                    read.round3_repeat_size = r2
                    read.round3_status = READ_TOO_LONG
                    continue
            reads.append(seq); kmin.append(lo); kmax.append(hi); rr.append(g); owners.append(read)
    if not reads:
        return None
    out = scorer(regions, reads, np.array(kmin, np.int32), np.array(kmax, np.int32),
                 read_region=np.array(rr, np.int32), sc=scoring, device=device, per_candidate=False)
    for i, read in enumerate(owners):
        st = int(out["status"][i])
        read.round3_status = st
        read.round3_best_score = int(out["best_score"][i])
        if st == _capi.READ_OK:                                  # :430-431, np.mean of the tied k
            read.round3_repeat_size = np.float64(out["sum_k"][i]) / np.float64(out["n_ties"][i])
        elif st == _capi.READ_FALLBACK:                          # :432-433
            read.round3_repeat_size = read.round2_repeat_size
        # NO_RECORD (:421 empty PAF) and SKIPPED (:460) leave round3_repeat_size untouched
    return out


def round3_estimation(data_type, fast_mode, repeat_region, num_cpu=1, device=0, scoring=None,
                      scorer=None):
    """Drop-in for nanoRepeat_bam.round3_estimation (nanoRepeat_bam.py:446-450)."""
    round3_estimation_regions(data_type, fast_mode, [repeat_region], num_cpu, device, scoring, scorer)


def output_repeat_size_1d(repeat_region):
    """`<out_prefix>.repeat_size.txt` -- the parity artefact (split_alleles.py:536-558)."""
    lines = [f"##Repeat_Region={repeat_region.to_unique_id()}\n", "#Read_Name\tRepeat_Size\n"]
    for read_name in repeat_region.read_dict:
        repeat_size = repeat_region.read_dict[read_name].round3_repeat_size
        if repeat_size is not None:
            lines.append(f"{read_name}\t{repeat_size:.1f}\n")
    text = "".join(lines)
    from . import phasing
    phasing.record_repeat_sizes(repeat_region)
    if not repeat_region.no_details and repeat_region.out_prefix:
        with open(f"{repeat_region.out_prefix}.repeat_size.txt", "w") as f:
            f.write(text)
    return text

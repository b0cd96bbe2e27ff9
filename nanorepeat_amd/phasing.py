"""Step 4 of the reference pipeline: phasing reads into alleles with a Gaussian mixture and the
result files (split_alleles.py:82-534, nanoRepeat_bam.py:502-574,
nanoRepeat_joint.py:675-747).  Host-side statistics on a few hundred numbers per region --
there is no GPU work here; scikit-learn does the fitting exactly as in the reference.

One implementation serves both the 1D and the joint (2D) mode: the reference's pairs of
`*_1d` / `*_2d` functions differ only in the number of columns.  What is mirrored:

* outliers: reads outside mean +- 3 sd on any axis are left out (lower cut clamped at 0);
* the mixture is fitted on a SIMULATED sample: every kept size repeated 100 times with
  gaussian noise of sd `error_rate * (10 + size)`; the real sizes are only classified;
* model order: fit n = first_n, first_n+1, ... components (diag covariance, n_init=10) and stop
  at the first n where two components overlap -- their central intervals
  [isf(1-o), isf(o)] (sd floored at 1) intersect on EVERY axis -- then refit with n-1;
  `first_n` is 2 in 1D and 1 in 2D like the reference (so 1D never stops at n=1);
* per read: label = argmax posterior; LOW confidence if the posterior < 0.95 or the size is
  outside mean +- 2 sd of its component on any axis;
* alleles sorted by read count, empty ones dropped; optional removal of small components while
  there are more than `ploidy` (smallest.num_reads * 1.5 <= the ploidy-th largest); the joint
  mode then refits from scratch on the surviving reads; final order by the first axis' mean.

The reference is unseeded (global `random` + numpy's global RandomState).  `seed=None` keeps
that behaviour -- seed the two globals and the draws match the reference's; `seed=int` uses
private generators started from that seed, which yield the same streams.
"""
import math
import os
import random

import numpy as np

from . import io as nr_io

PROBABILITY_CUTOFF = 0.95
COV_TYPE = "diag"
SIM_COPIES = 100


def data_type_error_rate(data_type):
    """nanoRepeat_bam.py:691-703.  The reference's first test is `data_type == 'ont' or 'clr'`,
    which is always true, so every accepted data type gets 0.07; kept for drop-in parity."""
    if data_type not in ("ont", "ont_sup", "ont_q20", "clr", "hifi"):
        raise ValueError(f"unknown data type: {data_type}")
    return 0.07


class Allele:
    """split_alleles.py:52-71."""

    def __init__(self):
        self.gmm_mean1 = self.gmm_mean2 = None
        self.gmm_sd1 = self.gmm_sd2 = None
        self.gmm_min1 = self.gmm_min2 = self.gmm_max1 = self.gmm_max2 = None
        self.readname_list = []
        self.repeat1_size_list = []
        self.repeat2_size_list = []
        self.repeat1_median_size = None
        self.repeat2_median_size = None
        self.probability_list = []
        self.confidence_list = []
        self.num_reads = None
        self.allele_frequency = None

    def _axis(self, d):
        sizes = self.repeat1_size_list if d == 0 else self.repeat2_size_list
        lo = self.gmm_min1 if d == 0 else self.gmm_min2
        hi = self.gmm_max1 if d == 0 else self.gmm_max2
        return sizes, lo, hi


class Readinfo:
    """split_alleles.py:73-80."""

    def __init__(self, readname):
        self.readname = readname
        self.label = -1
        self.repeat_size1 = -1
        self.repeat_size2 = -1
        self.confidence = 1


class QuantifiedAllele:
    def __init__(self):
        self.num_supp_reads = "*"
        self.repeat_size1 = "*"
        self.repeat_size2 = "*"


class QuantifiedRead:
    def __init__(self, read_name="*"):
        self.read_name = read_name
        self.repeat_size1 = -1
        self.repeat_size2 = -1
        self.allele_id = -1
        self.phasing_confidence = -1


class Result:
    """repeat_region.py:72-113: what ends up in one row of `<prefix>.NanoRepeat_output.tsv`."""

    def __init__(self):
        self.quantified_allele_list = []
        self.quantified_read_dict = dict()
        self.num_alleles = None

    def allele_summary(self):
        return "Allele_Repeat_Size;Allele_Num_Support_Reads" + "".join(
            f"|{a.repeat_size1};{a.num_supp_reads}" for a in self.quantified_allele_list)

    def read_summary(self):
        return "Read_Name;Read_Repeat_Size;Read_Allele_ID;PhasingConfidence" + "".join(
            f"|{r.read_name};{r.repeat_size1};{r.allele_id};{r.phasing_confidence}"
            for r in self.quantified_read_dict.values())

    def max_repeat_size1(self):
        return max((a.repeat_size1 for a in self.quantified_allele_list), default=-1)

    def min_repeat_size1(self):
        return min((a.repeat_size1 for a in self.quantified_allele_list), default=-1)


def results_of(repeat_region):
    if getattr(repeat_region, "results", None) is None:
        repeat_region.results = Result()
    return repeat_region.results


def record_repeat_sizes(repeat_region):
    """The bookkeeping half of output_repeat_size_1d (split_alleles.py:545-553): every read with
    a round-3 size gets a row in the final table, phased or not."""
    res = results_of(repeat_region)
    for read_name, read in repeat_region.read_dict.items():
        if read.round3_repeat_size is not None and read_name not in res.quantified_read_dict:
            q = QuantifiedRead(read_name)
            q.repeat_size1 = read.round3_repeat_size
            res.quantified_read_dict[read_name] = q


def final_output_row(repeat_region):
    """repeat_region.py:186-191 (`get_final_output`)."""
    res = results_of(repeat_region)
    start = max(0, repeat_region.start_pos)
    row = (f"{repeat_region.chrom}\t{start}\t{repeat_region.end_pos}\t{repeat_region.repeat_unit_seq}\t"
           f"{len(res.quantified_allele_list)}\t{res.max_repeat_size1()}\t{res.min_repeat_size1()}\t"
           f"{res.allele_summary()}\t{res.read_summary()}\n")
    repeat_region.final_output = row
    return row


def outfile_prefix(repeat_region):
    """repeat_region.py:178-184."""
    seq = repeat_region.repeat_unit_seq
    if len(seq) >= 30:
        seq = seq[0:20] + "...." + seq[-6:]
    return f"{repeat_region.chrom}-{repeat_region.start_pos}-{repeat_region.end_pos}-{seq}"


# ---------------------------------------------------------------------------------------------
# statistics
# ---------------------------------------------------------------------------------------------
def get_outlier_cutoff_from_list(repeat_count_list):
    """split_alleles.py:98-112."""
    if len(repeat_count_list) == 0:
        raise ValueError("no repeat sizes to analyse")
    mean, std = np.mean(repeat_count_list), np.std(repeat_count_list)
    return max(0, mean - 3 * std), mean + 3 * std


def remove_outlier_reads(count_dict, dimension):
    """split_alleles.py:124-154.  count_dict: {read: size} (1D) or {read: (size1, size2)} (2D).
    Returns the kept read names and their sizes flattened row-major."""
    rows = [(name, (v,) if dimension == 1 else tuple(v)) for name, v in count_dict.items()]
    cuts = [get_outlier_cutoff_from_list([r[1][d] for r in rows]) for d in range(dimension)]
    names, flat = [], []
    for name, v in rows:
        if any(v[d] < cuts[d][0] or v[d] > cuts[d][1] for d in range(dimension)):
            continue
        names.append(name)
        flat.extend(v)
    return names, flat


def simulate_reads(read_repeat_count_list, error_rate, rng=None):
    """split_alleles.py:82-88: the list 100 times over, each value with its own gaussian error."""
    gauss = (rng or random).gauss
    out = list(read_repeat_count_list) * SIM_COPIES
    for i, v in enumerate(out):
        out[i] = v + gauss(0, error_rate * (10 + v))
    return out


def _central_interval(mean, cov, overlap):
    from scipy.stats import norm
    sd = max(1.0, math.sqrt(cov))
    return float(norm.isf(1.0 - overlap, mean, sd)), float(norm.isf(overlap, mean, sd))


def interval_has_overlap(interval1, interval2):
    """split_alleles.py:90-96 (touching intervals count as overlapping)."""
    return max(interval1[0], interval2[0]) - min(interval1[1], interval2[1]) <= 0


def _fit(X, n, random_state):
    from sklearn.mixture import GaussianMixture
    return GaussianMixture(n_components=n, covariance_type=COV_TYPE, n_init=10,
                           random_state=random_state).fit(X)


def auto_gmm(X, max_num_components, max_mutual_overlap, random_state=None):
    """split_alleles.py:171-240 for either dimension (X.shape[1])."""
    dimension = X.shape[1]
    for n in range(2 if dimension == 1 else 1, max_num_components + 1):
        gmm = _fit(X, n, random_state)
        iv = [[_central_interval(gmm.means_[c][d], gmm.covariances_[c][d], max_mutual_overlap)
               for d in range(dimension)] for c in range(n)]
        for i in range(n):
            for j in range(i + 1, n):
                if all(interval_has_overlap(iv[i][d], iv[j][d]) for d in range(dimension)):
                    return n - 1, _fit(X, n - 1, random_state)
    return max_num_components, _fit(X, max_num_components, random_state)


def create_allele_list(best_n_components, final_gmm, readname_list, read_repeat_count_array, count_dict,
                       probability_cutoff=PROBABILITY_CUTOFF):
    """split_alleles.py:242-355."""
    dimension = read_repeat_count_array.shape[1]
    labels = final_gmm.predict(read_repeat_count_array)
    proba = final_gmm.predict_proba(read_repeat_count_array)
    assert len(labels) == len(readname_list)
    alleles = []
    for c in range(best_n_components):
        a = Allele()
        a.gmm_mean1 = float(final_gmm.means_[c][0])
        a.gmm_sd1 = math.sqrt(final_gmm.covariances_[c][0])
        if dimension == 2:
            a.gmm_mean2 = float(final_gmm.means_[c][1])
            a.gmm_sd2 = math.sqrt(final_gmm.covariances_[c][1])
        alleles.append(a)
    for i, readname in enumerate(readname_list):
        a = alleles[labels[i]]
        v = count_dict[readname]
        a.readname_list.append(readname)
        a.repeat1_size_list.append(v if dimension == 1 else v[0])
        if dimension == 2:
            a.repeat2_size_list.append(v[1])
        a.probability_list.append(proba[i][labels[i]])
    for a in alleles:
        a.num_reads = len(a.readname_list)
        if a.num_reads == 0:
            a.repeat1_median_size = 0
            a.gmm_min1 = a.gmm_max1 = 0
            if dimension == 2:
                a.repeat2_median_size = 0
                a.gmm_min2 = a.gmm_max2 = 0
            continue
        a.repeat1_median_size = int(np.median(a.repeat1_size_list) + 0.5)
        a.gmm_min1, a.gmm_max1 = a.gmm_mean1 - 2 * a.gmm_sd1, a.gmm_mean1 + 2 * a.gmm_sd1
        if dimension == 2:
            a.repeat2_median_size = int(np.median(a.repeat2_size_list) + 0.5)
            a.gmm_min2, a.gmm_max2 = a.gmm_mean2 - 2 * a.gmm_sd2, a.gmm_mean2 + 2 * a.gmm_sd2
    for a in alleles:
        a.confidence_list = []
        for i in range(a.num_reads):
            low = a.probability_list[i] < probability_cutoff
            for d in range(dimension):
                sizes, lo, hi = a._axis(d)
                low = low or sizes[i] < lo or sizes[i] > hi
            a.confidence_list.append("LOW" if low else "HIGH")
    alleles.sort(key=lambda a: a.num_reads)
    while alleles[0].num_reads == 0:
        alleles.pop(0)
    return alleles


def remove_noisy_alleles(allele_list, ploidy):
    """nanoRepeat_bam.py:502-514 / nanoRepeat_joint.py:675-684."""
    allele_list.sort(key=lambda a: a.num_reads)
    num_removed_reads = 0
    while len(allele_list) > ploidy and len(allele_list) >= 2:
        if allele_list[0].num_reads * 1.5 > allele_list[-ploidy].num_reads:
            break
        num_removed_reads += allele_list.pop(0).num_reads
    return allele_list, num_removed_reads


def create_readinfo_dict_from_allele_list(allele_list, dimension):
    """split_alleles.py:359-378."""
    if dimension not in (1, 2):
        raise ValueError("dimension must be 1 or 2")
    info = dict()
    for label, a in enumerate(allele_list):
        for i, readname in enumerate(a.readname_list):
            r = Readinfo(readname)
            r.label = label
            r.repeat_size1 = a.repeat1_size_list[i]
            if dimension == 2:
                r.repeat_size2 = a.repeat2_size_list[i]
            r.confidence = a.confidence_list[i]
            info[readname] = r
    return info


def _generators(seed):
    if seed is None:
        return None, None                      # the two global generators, like the reference
    return random.Random(seed), np.random.RandomState(seed)


def phase(count_dict, dimension, ploidy, error_rate, max_mutual_overlap, max_num_components,
          remove_noisy_reads, seed=None, _gens=None):
    """The statistical core shared by both drivers.  Returns (allele_list, num_removed_reads,
    final_gmm), or None when there are too few reads."""
    if ploidy < 1:
        raise ValueError("ploidy must be >= 1")
    py_rng, np_rng = _gens if _gens is not None else _generators(seed)
    names, flat = remove_outlier_reads(count_dict, dimension)
    real = np.array(flat).reshape(-1, dimension)
    simulated = np.array(simulate_reads(flat, error_rate, py_rng)).reshape(-1, dimension)
    n, gmm = auto_gmm(simulated, max_num_components, max_mutual_overlap, np_rng)
    alleles = create_allele_list(n, gmm, names, real, count_dict)
    num_removed = 0
    if remove_noisy_reads and len(alleles) > ploidy:
        alleles, num_removed = remove_noisy_alleles(alleles, ploidy)
        if dimension == 2:
            # the joint mode starts over on the surviving reads (nanoRepeat_joint.py:686-696) and
            # reports 0 removed reads afterwards (`num_removed_reads = 0` at :723)
            kept = {name: (a.repeat1_size_list[i], a.repeat2_size_list[i])
                    for a in alleles for i, name in enumerate(a.readname_list)}
            if len(kept) < ploidy or len(kept) == 1:
                return None
            return phase(kept, 2, ploidy, error_rate, max_mutual_overlap, max_num_components, False,
                         _gens=(py_rng, np_rng))
    alleles.sort(key=lambda a: a.gmm_mean1)
    return alleles, num_removed, gmm


# ---------------------------------------------------------------------------------------------
# result files
# ---------------------------------------------------------------------------------------------
def phased_reads_text(allele_list, header, dimension):
    """split_alleles.py:380-437."""
    out = [header]
    for label, a in enumerate(allele_list):
        for i, readname in enumerate(a.readname_list):
            row = f"{readname}\t{label + 1}\t{a.confidence_list[i]}\t{a.repeat1_size_list[i]:.1f}"
            if dimension == 2:
                row += f"\t{a.repeat2_size_list[i]:.1f}"
            out.append(row + "\n")
    return "".join(out)


def output_phased_fastq(in_fastq_file, readinfo_dict, num_alleles, out_prefix):
    """split_alleles.py:440-481: one FASTQ per allele with its HIGH-confidence reads, records
    copied verbatim in input order."""
    files = [open(f"{out_prefix}.allele{label + 1}.fastq", "w") for label in range(num_alleles)]
    try:
        with nr_io.gzopen(in_fastq_file, "rt") as f:
            while True:
                rec = [f.readline() for _ in range(4)]
                if not all(rec):
                    break
                readname = rec[0].strip().split()[0][1:]
                info = readinfo_dict.get(readname)
                if info is None or info.confidence != "HIGH":
                    continue
                files[info.label].write("".join(rec))
    finally:
        for f in files:
            f.close()


def region_count_dict(repeat_region):
    """{read: round-3 size} of the reads of a region that have one (nanoRepeat_bam.py:527-530)."""
    return {name: r.round3_repeat_size for name, r in repeat_region.read_dict.items()
            if r.round3_repeat_size is not None}


def phase_1d_job(args):
    """The mixture fit of one region without the region object (picklable in, picklable out): what
    `pipeline.phase_regions` hands to its worker processes.  Returns (alleles, num_removed) or None."""
    count_dict, ploidy, error_rate, max_mutual_overlap, max_num_components, remove_noisy_reads, seed = args
    if len(count_dict) < 2:
        return None
    alleles, num_removed, _ = phase(count_dict, 1, ploidy, error_rate, max_mutual_overlap,
                                    max_num_components, remove_noisy_reads, seed)
    return alleles, num_removed


def split_allele_using_gmm_1d(repeat_region, ploidy, error_rate, max_mutual_overlap, max_num_components,
                              remove_noisy_reads, seed=None, fitted=None):
    """Drop-in for nanoRepeat_bam.split_allele_using_gmm_1d (nanoRepeat_bam.py:517-574) minus the
    plots: fills `repeat_region.results`, writes `<out_prefix>.phased_reads.txt`, `.summary.txt`
    (unless `no_details`) and `.alleleN.fastq` (when `region_fq_file` is set).  Returns the allele
    list, or None when fewer than 2 reads have a size.  `fitted` = the result of `phase_1d_job` for
    this region when the fit was done elsewhere (a worker process)."""
    if ploidy < 1:
        raise ValueError("ploidy must be >= 1")
    if fitted is None:
        fitted = phase_1d_job((region_count_dict(repeat_region), ploidy, error_rate, max_mutual_overlap,
                               max_num_components, remove_noisy_reads, seed))
    if fitted is None:
        return None
    alleles, num_removed = fitted
    res = results_of(repeat_region)
    for label, a in enumerate(alleles):
        for i, readname in enumerate(a.readname_list):
            q = res.quantified_read_dict.setdefault(readname, QuantifiedRead(readname))
            q.repeat_size1 = a.repeat1_size_list[i]
            q.allele_id = label + 1
            q.phasing_confidence = a.confidence_list[i]
        qa = QuantifiedAllele()
        qa.repeat_size1, qa.num_supp_reads = a.repeat1_median_size, a.num_reads
        res.quantified_allele_list.append(qa)
    res.num_alleles = len(alleles)

    details = not repeat_region.no_details and repeat_region.out_prefix
    if details:
        header = f"##RepeatRegion={repeat_region.to_unique_id()}\n#Read_Name\tAllele_ID\tPhasing_Confidence\tRepeat_Size\n"
        with open(repeat_region.out_prefix + ".phased_reads.txt", "w") as f:
            f.write(phased_reads_text(alleles, header, 1))
    fq = getattr(repeat_region, "region_fq_file", None)
    if fq and repeat_region.out_prefix:
        output_phased_fastq(fq, create_readinfo_dict_from_allele_list(alleles, 1), len(alleles),
                            repeat_region.out_prefix)
    if details:
        out_file = repeat_region.out_prefix + ".summary.txt"
        text = (f"Summary_file={os.path.split(out_file)[1]}\tRepeat_Region={repeat_region.to_unique_id()}"
                f"\tMethod=GMM\tNum_Alleles={len(alleles)}\tNum_Removed_Reads={num_removed}")
        for label, a in enumerate(alleles):
            text += f"\tAllele{label + 1}_Num_Reads={a.num_reads}\tAllele{label + 1}_Repeat_Size={a.repeat1_median_size}"
        with open(out_file, "w") as f:
            f.write(text + "\n")
    return alleles


def phase_2d_job(args):
    """The 2D mixture fit as a picklable job (see phase_1d_job): (alleles, components of the final
    mixture) or None."""
    count_dict, ploidy, error_rate, max_mutual_overlap, max_num_components, remove_noisy_reads, seed = args
    if len(count_dict) < ploidy or len(count_dict) == 1:
        return None
    got = phase(count_dict, 2, ploidy, error_rate, max_mutual_overlap, max_num_components, remove_noisy_reads, seed)
    if got is None:
        return None
    return got[0], int(got[2].n_components)


def run_job(job):
    """("1d" | "2d", args) -> result: the unit of work of the worker processes."""
    kind, args = job
    return phase_1d_job(args) if kind == "1d" else phase_2d_job(args)


def split_alleles_using_gmm_2d(ploidy, error_rate, max_mutual_overlap, remove_noisy_reads, max_num_components,
                               repeat1, repeat2, read_repeat_joint_count_dict, num_removed_reads, in_fastq_file,
                               out_prefix, seed=None, fitted=None):
    """Drop-in for nanoRepeat_joint.split_alleles_using_gmm_2d (nanoRepeat_joint.py:699-747) minus
    the plots.  `num_removed_reads` is accepted and ignored like in the reference (it is reset to
    0 before use).  Returns the allele list or None (too few reads).  `fitted` = the result of
    `phase_2d_job` when the fit was done elsewhere (a worker process)."""
    if ploidy < 1:
        raise ValueError("ploidy must be >= 1")
    if fitted is None:
        fitted = phase_2d_job((read_repeat_joint_count_dict, ploidy, error_rate, max_mutual_overlap,
                               max_num_components, remove_noisy_reads, seed))
    if fitted is None:
        return None
    alleles, n_components = fitted
    id1, id2 = repeat1.repeat_id, repeat2.repeat_id
    header = (f"##Input_FASTQ={in_fastq_file}\n"
              f"#Read_Name\tAllele_ID\tPhasing_Confidence\t{id1}.Repeat_Size\t{id2}.Repeat_Size\n")
    with open(out_prefix + ".phased_reads.txt", "w") as f:
        f.write(phased_reads_text(alleles, header, 2))
    if in_fastq_file and os.path.exists(in_fastq_file):
        # one file per mixture component, also for components that ended up without reads
        output_phased_fastq(in_fastq_file, create_readinfo_dict_from_allele_list(alleles, 2),
                            max(n_components, len(alleles)), out_prefix)
    text = f"Input_FASTQ\t{in_fastq_file}\nMethod\t2D-GMM\nNum_Alleles\t{len(alleles)}\nNum_Removed_Reads\t0\n"
    for label, a in enumerate(alleles):
        text += (f"Allele{label + 1}_Num_Reads\t{a.num_reads}\n"
                 f"Allele{label + 1}_{id1}.Repeat_Size\t{a.repeat1_median_size}\n"
                 f"Allele{label + 1}_{id2}.Repeat_Size\t{a.repeat2_median_size}\n")
    with open(out_prefix + ".summary.txt", "w") as f:
        f.write(text)
    return alleles

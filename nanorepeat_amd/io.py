"""Input ingestion for the rows around the path: repeat-region BED, reference
FASTA, region FASTQ, and flank extraction.  Mirrors repeat_region.py:195-204, tk.py:68-75,
130-158 and nanoRepeat_bam.py:76-137 with the reference's names; errors raise instead of
calling sys.exit().  BAM input needs pysam, which is absent offline: reads enter as FASTQ/FASTA
or as a {name: sequence} dict."""
import gzip

from .round3 import RepeatRegion

MIN_ANCHOR_LEN = 10


def gzopen(in_file, mode="rt"):
    """tk.gzopen (tk.py:68-75): transparent .gz."""
    if str(in_file).endswith(".gz"):
        return gzip.open(in_file, mode)
    return open(in_file, mode)


def read_repeat_region_file(repeat_region_file, no_details=False):
    """repeat_region.py:195-204: one RepeatRegion per line (also the last line without a newline,
    and CRLF files like the reference's example_data/HTT_repeat_region.bed)."""
    with open(repeat_region_file, "r") as f:
        lines = list(f)
    return [RepeatRegion(line, no_details) for line in lines]


def fasta_file2dict(fasta_file):
    """tk.fasta_file2dict (tk.py:130-158): name = first word of the header, sequence upper-cased."""
    fasta_dict = dict()
    curr_name, chunks = "", []
    with gzopen(fasta_file) as fp:
        for line in fp:
            line = line.strip()
            if not line:
                continue
            if line[0] == ">":
                if chunks and curr_name:
                    if curr_name in fasta_dict:
                        raise ValueError(f"duplicate sequence name in {fasta_file}: {curr_name}")
                    fasta_dict[curr_name] = "".join(chunks)
                curr_name, chunks = line[1:].split()[0], []
                continue
            chunks.append(line.upper())
    if chunks and curr_name:
        fasta_dict[curr_name] = "".join(chunks)
    return fasta_dict


def read_fastq(fastq_file):
    """{read_name: sequence} in file order (4-line records; name = first word of the header)."""
    reads = dict()
    with gzopen(fastq_file) as fp:
        while True:
            l1, l2, l3, l4 = fp.readline(), fp.readline(), fp.readline(), fp.readline()
            if not l1 or not l2 or not l3 or not l4:
                break
            reads[l1.strip()[1:].split()[0]] = l2.strip()
    return reads


def fastq_file_to_dict(in_fastq_file):
    """nanoRepeat_joint.py:652-673: {read_name: the record's four lines as one string}."""
    fastq_dict = dict()
    with gzopen(in_fastq_file) as fp:
        while True:
            rec = [fp.readline() for _ in range(4)]
            if not all(rec):
                break
            fastq_dict[rec[0].strip().split()[0][1:]] = "".join(rec)
    return fastq_dict


def read_one_chr_from_fasta_file(fasta_file, target_chr):
    """tk.read_one_chr_from_fasta_file (tk.py:193-230): the first record with that exact name,
    upper-cased; '' when absent."""
    chunks, reading = [], False
    with gzopen(fasta_file) as fp:
        for line in fp:
            line = line.strip()
            if not line:
                continue
            if line[0] == ">":
                if reading and chunks:
                    break
                reading = line[1:].split()[0] == target_chr
                continue
            if reading:
                chunks.append(line.upper())
    return "".join(chunks)


def _lookup_chromosome(ref_fasta_dict, name):
    """The chromosome under its BED name, or under the other naming convention ("chr4" <-> "4")."""
    other = name[3:] if name.startswith("chr") else "chr" + name
    for key in (name, other):
        if key in ref_fasta_dict:
            return ref_fasta_dict[key]
    raise KeyError(f"chromosome {name} of the repeat region bed file is not in the reference fasta file")


def _interval_problem(start, end, chr_len):
    """Why [start, end) is not a usable repeat interval on a chromosome of chr_len bases (None: it is).
    The reference's four checks (nanoRepeat_bam.py:91-110), which it answers with sys.exit()."""
    for bad, why in ((start > chr_len, f"the repeat start position is larger than chromosome length: {start}"),
                     (start < 0, "the repeat start position < 0"),
                     (end > chr_len + 1, f"the repeat end position is larger than chromosome length: {end}"),
                     (end < start, "end position is smaller than start position")):
        if bad:
            return why
    return None


def extract_ref_sequence(ref_fasta_dict, repeat_region, anchor_len=1000):
    """Mirror of nanoRepeat_bam.py:76-137: sets the region's two anchors (at most anchor_len bases each,
    clamped to the chromosome) and the reference slice between them; raises where the reference exits."""
    rr = repeat_region
    rr.anchor_len = max(anchor_len, MIN_ANCHOR_LEN)
    chrom = _lookup_chromosome(ref_fasta_dict, rr.chrom)
    problem = _interval_problem(rr.start_pos, rr.end_pos, len(chrom))
    if problem:
        raise ValueError(problem)
    left = chrom[max(0, rr.start_pos - rr.anchor_len):rr.start_pos]
    right = chrom[rr.end_pos:min(len(chrom), rr.end_pos + rr.anchor_len)]
    if not left and not right:
        raise ValueError("there is no flanking sequence around the repeat region")
    if max(len(left), len(right)) < MIN_ANCHOR_LEN:
        raise ValueError(f"both left and right flanking sequences are less than {MIN_ANCHOR_LEN} bp")
    rr.left_anchor_seq, rr.left_anchor_len = left, len(left)
    rr.right_anchor_seq, rr.right_anchor_len = right, len(right)
    rr.mid_ref_seq = chrom[rr.start_pos:rr.end_pos]


def edit_distance(a, b):
    """Levenshtein distance (the reference imports the Levenshtein package, absent offline)."""
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def check_repeat_motif_in_ref(repeat_region):
    """nanoRepeat_bam.py:139-154: the reference locus must look like the motif (edit distance of
    the pure repeat to the reference slice at most a quarter of the shorter one)."""
    mid, unit = repeat_region.mid_ref_seq, repeat_region.repeat_unit_seq
    pure = unit * int(len(mid) / len(unit))
    repeat_region.ref_has_issue = edit_distance(pure, mid) * 4 > min(len(pure), len(mid))
    return not repeat_region.ref_has_issue

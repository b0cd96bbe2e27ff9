"""Steps 1-4 of quantify1repeat_from_bam (nanoRepeat_bam.py:614-686) for one or many regions,
from reads already extracted for the region: anchors -> core -> rounds 1-2 -> round 3 -> the
`repeat_size.txt` text -> GMM phasing -> one row of `NanoRepeat_output.tsv` per region.
`quantify_from_bam` and `quantify_joint` are the two commands from files to files."""
from . import upstream, round3, phasing, joint, io as nr_io


def quantify_regions(repeat_regions, reads_by_region, data_type="ont", fast_mode=False, num_cpu=1,
                     device=0, scoring=None, aligner=None, scorer=None):
    """repeat_regions: RepeatRegion objects with anchors set (io.extract_ref_sequence);
    reads_by_region: one {read_name: sequence} dict per region.  Every step runs for all regions in
    one C-ABI call (per ~256 M bases): a call costs a few milliseconds however small it is."""
    upstream.find_anchor_locations_in_reads_many(data_type, repeat_regions, reads_by_region, num_cpu, device=device,
                                                 scoring=scoring, aligner=aligner)
    for region, reads in zip(repeat_regions, reads_by_region):
        upstream.make_core_seq(region, reads)
    upstream.round1_and_round2_estimation_many(data_type, repeat_regions, num_cpu, device=device, scoring=scoring,
                                               aligner=aligner)
    round3.round3_estimation_regions(data_type, fast_mode, repeat_regions, num_cpu, device, scoring, scorer)
    report_unscored_reads(repeat_regions)
    return [round3.output_repeat_size_1d(region) for region in repeat_regions]


def report_unscored_reads(repeat_regions, stream=None):
    """Reads that did not get a round-3 size of their own must not blend in silently: reads beyond the C
    ABI's length limits (left out of steps 1-2, or kept at their round-2 size in step 3) are listed per
    region on `region.skipped_reads` and counted on stderr.  Returns the number of such reads."""
    import sys
    stream = stream or sys.stderr
    total = 0
    for region in repeat_regions:
        skipped = dict(getattr(region, "skipped_reads", None) or {})
        for name, read in region.read_dict.items():
            if getattr(read, "round3_status", None) == round3.READ_TOO_LONG:
                skipped[name] = "core or template beyond the scorer's limits: kept at its round-2 size"
        region.skipped_reads = skipped
        if skipped:
            total += len(skipped)
            some = ", ".join(list(skipped)[:5]) + (" ..." if len(skipped) > 5 else "")
            print(f"NOTICE: {region.to_unique_id()}: {len(skipped)} read(s) beyond the length limits were not "
                  f"scored in full ({some})", file=stream)
    return total


def _fit_in_worker_processes(jobs, n_jobs):
    """phasing.run_job over `jobs` in n_jobs fresh interpreters (`python -m
    nanorepeat_amd._phase_worker`): they get only the sizes, never import the caller's main module
    and never open the GPU."""
    import os, pickle, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    shares = [list(range(w, len(jobs), n_jobs)) for w in range(n_jobs)]
    # start every interpreter first (they import scikit-learn side by side), then hand out the work
    procs = [subprocess.Popen([sys.executable, "-m", "nanorepeat_amd._phase_worker"], stdin=subprocess.PIPE,
                              stdout=subprocess.PIPE, env=env) for _ in shares]
    for share, p in zip(shares, procs):
        p.stdin.write(pickle.dumps([jobs[i] for i in share]))
        p.stdin.close()
    fitted = [None] * len(jobs)
    for share, p in zip(shares, procs):
        data = p.stdout.read()
        if p.wait() != 0:
            raise RuntimeError("a phasing worker process failed")
        for i, res in zip(share, pickle.loads(data)):
            fitted[i] = res
    return fitted


def phase_regions(repeat_regions, data_type="ont", ploidy=2, max_mutual_overlap=0.15, max_num_components=-1,
                  remove_noisy_reads=False, seed=None, out_tsv_file=None, n_jobs=None):
    """Step 4 for every region (nanoRepeat_bam.py:683-684) and the final table (:737-743); defaults
    are the CLI's (nanoRepeat.py:121-129,159-160).  With a seed, region i uses seed + i.  The mixture
    fits -- by far the longest step of the whole command -- run in up to 16 worker processes like
    the reference's region workers (nanoRepeat_bam.py:712-724); the workers are fresh interpreters
    that get only the sizes and never touch the GPU.  n_jobs=1 keeps everything in this process."""
    import os
    if max_num_components == -1:
        max_num_components = ploidy + 20
    error_rate = phasing.data_type_error_rate(data_type)
    jobs = [(phasing.region_count_dict(region), ploidy, error_rate, max_mutual_overlap, max_num_components,
             remove_noisy_reads, None if seed is None else seed + i) for i, region in enumerate(repeat_regions)]
    if n_jobs is None:
        n_jobs = min(16, os.cpu_count() or 1, max(1, sum(len(j[0]) >= 2 for j in jobs)))
    if n_jobs > 1:
        fitted = _fit_in_worker_processes([("1d", j) for j in jobs], n_jobs)
    else:
        fitted = [phasing.phase_1d_job(j) for j in jobs]
    rows = []
    for region, job, fit in zip(repeat_regions, jobs, fitted):
        if fit is not None:
            phasing.split_allele_using_gmm_1d(region, ploidy, error_rate, max_mutual_overlap, max_num_components,
                                              remove_noisy_reads, fitted=fit)
        rows.append(phasing.final_output_row(region))
    if out_tsv_file:
        with open(out_tsv_file, "w") as f:
            f.write("".join(rows))
    return rows


def quantify_joint(in_fq, ref_fasta, repeat1_string, repeat2_string, out_prefix, data_type="ont", num_threads=1,
                   ploidy=2, error_rate=0.1, max_mutual_overlap=0.1, remove_noisy_reads=False,
                   max_num_components=-1, device=0, scoring=None, seed=None, phase_in_worker=True, **engines):
    """The joint (2D) command from files to files (nanoRepeat_joint.py:160-232): round 1 ->
    grid rounds 2/3 -> `<out_prefix>.repeat_size.txt` -> 2D GMM phasing -> `.phased_reads.txt`,
    `.summary.txt`, `.alleleN.fastq`.  `engines` may carry aligner / cigar_aligner / scorer
    stand-ins (tests).  Returns (RepeatSize, allele list or None)."""
    if max_num_components == -1:
        max_num_components = ploidy + 20
    fastq_dict = nr_io.fastq_file_to_dict(in_fq)
    if len(fastq_dict) < ploidy:
        raise ValueError(f"not enough reads for analysis: ploidy {ploidy}, {len(fastq_dict)} reads in {in_fq}")
    repeat1 = joint.Repeat().init_from_string(repeat1_string)
    repeat2 = joint.Repeat().init_from_string(repeat2_string)
    if repeat1.chrom != repeat2.chrom:
        raise ValueError("joint quantification only works with two nearby repeats on one chromosome")
    if repeat1.start > repeat2.start:
        repeat1, repeat2 = repeat2, repeat1
    repeat1.max_size += 10
    repeat2.max_size += 10
    if repeat1.end + 100 < repeat2.start:
        raise ValueError("joint quantification only works with two nearby repeats (distance < 100 bp)")
    repeat_chrom_seq = nr_io.read_one_chr_from_fasta_file(ref_fasta, repeat1.chrom)
    if not repeat_chrom_seq:
        raise ValueError(f"ref_fasta file {ref_fasta} has no sequence named {repeat1.chrom}")
    initial_estimation = joint.initial_estimate_repeat_size(
        repeat_chrom_seq, fastq_dict, data_type, num_threads, repeat1, repeat2, 1000, device=device, scoring=scoring,
        aligner=engines.get("aligner"), cigar_aligner=engines.get("cigar_aligner"))
    final_estimation = joint.fine_tune_read_count(initial_estimation, fastq_dict, repeat_chrom_seq, repeat1, repeat2,
                                                  data_type, num_threads, None, device, scoring, engines.get("scorer"))
    joint_counts, _ = joint.output_repeat_size_2d(in_fq, repeat1.repeat_id, repeat2.repeat_id, out_prefix,
                                                  final_estimation.repeat1_count_dict,
                                                  final_estimation.repeat2_count_dict)
    # the fit runs in a fresh single-threaded interpreter: scikit-learn's small-matrix algebra is several
    # times slower with this process's BLAS/OpenMP thread pools (split_alleles.py:28-32 pins them to 1)
    job = ("2d", (joint_counts, ploidy, error_rate, max_mutual_overlap, max_num_components, remove_noisy_reads, seed))
    fitted = _fit_in_worker_processes([job], 1)[0] if phase_in_worker else None
    alleles = phasing.split_alleles_using_gmm_2d(ploidy, error_rate, max_mutual_overlap, remove_noisy_reads,
                                                 max_num_components, repeat1, repeat2, joint_counts, 0, in_fq,
                                                 out_prefix, seed=seed, fitted=fitted)
    return final_estimation, alleles


def quantify_from_bam(in_bam_file, ref_fasta, repeat_region_bed, out_prefix, data_type="ont", anchor_len=1000,
                      fast_mode=False, ploidy=2, max_mutual_overlap=0.15, max_num_components=-1,
                      remove_noisy_reads=False, no_check_repeat_motif_in_ref=False, no_details=False,
                      num_cpu=1, device=0, scoring=None, seed=None, **engines):
    """The BAM command from files to files (nanoRepeat_bam.py:614-751): for every region of the BED
    file, reads from the alignment file -> `<out_prefix>.details/<chr>/<region>.*` ->
    `<out_prefix>.NanoRepeat_output.tsv`.  The reference forks up to 16 workers, one region each;
    here steps 1-2 run region by region and step 3 for all regions in one GPU batch.  Regions
    without reads, or whose reference sequence fails the motif check, get their row with 0
    alleles like in the reference.  Returns the regions."""
    import os
    from . import bam as nr_bam
    regions = nr_io.read_repeat_region_file(repeat_region_bed, no_details)
    ref_fasta_dict = nr_io.fasta_file2dict(ref_fasta)
    live, reads_of = [], []
    for i, region in enumerate(regions):
        region.index = i
        chrom_dir = region.chrom if region.chrom[0:3].lower() == "chr" else "chr" + region.chrom
        out_dir = f"{out_prefix}.details/{chrom_dir}"
        os.makedirs(out_dir, exist_ok=True)
        region.out_prefix = f"{out_dir}/{phasing.outfile_prefix(region)}"
        region.region_fq_file = f"{region.out_prefix}.reads.fastq"
        n = nr_bam.extract_fastq_from_bam(in_bam_file, region, anchor_len, region.region_fq_file, ref_fasta)
        if n == 0:
            continue
        nr_io.extract_ref_sequence(ref_fasta_dict, region, anchor_len)
        if not no_check_repeat_motif_in_ref and not nr_io.check_repeat_motif_in_ref(region):
            continue
        live.append(region)
        reads_of.append(nr_io.read_fastq(region.region_fq_file))
    quantify_regions(live, reads_of, data_type, fast_mode, num_cpu, device, scoring,
                     engines.get("aligner"), engines.get("scorer"))
    phase_regions(live, data_type, ploidy, max_mutual_overlap, max_num_components, remove_noisy_reads, seed)
    with open(f"{out_prefix}.NanoRepeat_output.tsv", "w") as f:
        for region in regions:
            f.write(phasing.final_output_row(region))
    if no_details:
        import shutil
        shutil.rmtree(f"{out_prefix}.details", ignore_errors=True)
    return regions

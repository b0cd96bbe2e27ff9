"""Steps 1-3 of quantify1repeat_from_bam (nanoRepeat_bam.py:614-686) for one or many regions,
from reads already extracted for the region: anchors -> core -> rounds 1-2 -> round 3 -> the
`repeat_size.txt` text.  Step 4 (GMM phasing) and BAM extraction are outside this build."""
from . import upstream, round3


def quantify_regions(repeat_regions, reads_by_region, data_type="ont", fast_mode=False, num_cpu=1,
                     device=0, scoring=None, aligner=None, scorer=None):
    """repeat_regions: RepeatRegion objects with anchors set (io.extract_ref_sequence);
    reads_by_region: one {read_name: sequence} dict per region.  Steps 1-2 run per region (their
    alignments are few and long); step 3 -- the hot path -- runs for all regions in one batch."""
    for region, reads in zip(repeat_regions, reads_by_region):
        upstream.find_anchor_locations_in_reads(data_type, region, num_cpu, region_reads=reads, device=device,
                                                scoring=scoring, aligner=aligner)
        upstream.make_core_seq(region, reads)
        upstream.round1_and_round2_estimation(data_type, region, num_cpu, device=device, scoring=scoring,
                                              aligner=aligner)
    round3.round3_estimation_regions(data_type, fast_mode, repeat_regions, num_cpu, device, scoring, scorer)
    return [round3.output_repeat_size_1d(region) for region in repeat_regions]

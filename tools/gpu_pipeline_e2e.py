"""Wall time of steps 1-4 of the BAM command for many regions from reads in memory
(pipeline.quantify_regions + phase_regions): where the host time goes."""
import cProfile, io, json, pstats, sys, time
import numpy as np
sys.path.insert(0, '.')
from nanorepeat_amd import synth, pipeline, round3 as R3

def main():
    n_reg = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    rng = np.random.default_rng(3)
    regions, reads_by_region, truth = [], [], []
    for g in range(n_reg):
        unit = synth.rand_unit(rng, int(rng.integers(3, 7)))
        left, right = synth.rand_seq(rng, 1000), synth.rand_seq(rng, 1000)
        rr = R3.RepeatRegion(f"chr1\t{1000 + 5000 * g}\t{1000 + 5000 * g + 10 * len(unit)}\t{unit}")
        rr.left_anchor_seq, rr.right_anchor_seq = left, right
        rr.left_anchor_len = rr.right_anchor_len = 1000
        alleles = (int(rng.integers(10, 60)), int(rng.integers(60, 120)))
        reads = {}
        for i in range(n_reads):
            k = alleles[i % 2]
            fl, fr = int(rng.integers(400, 1000)), int(rng.integers(400, 1000))
            s = synth.apply_errors(rng, left[1000 - fl:] + unit * k + right[:fr], "ont_q20")
            reads[f"g{g}r{i}"] = synth.revcomp(s) if i % 3 == 0 else s
        regions.append(rr); reads_by_region.append(reads); truth.append(alleles)
    pipeline.quantify_regions(regions[:2], reads_by_region[:2], "ont_q20")            # warm-up (context, code objects)
    pr = cProfile.Profile(); pr.enable()
    t0 = time.time(); pipeline.quantify_regions(regions, reads_by_region, "ont_q20"); t_q = time.time() - t0
    pr.disable()
    t0 = time.time(); rows = pipeline.phase_regions(regions, "ont_q20", seed=1); t_p = time.time() - t0
    ok = sum(sorted((int(r.split("\t")[5]), int(r.split("\t")[6]))) == sorted((max(t), min(t))) or
             all(abs(x - y) <= 1 for x, y in zip(sorted((int(r.split("\t")[5]), int(r.split("\t")[6]))), sorted(t))) for r, t in zip(rows, truth))
    print(json.dumps({"regions": n_reg, "reads": n_reg * n_reads, "steps_1_to_3_s": round(t_q, 3), "phasing_s": round(t_p, 3), "regions_with_both_alleles_within_1": ok}))
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(14); print(s.getvalue()[:2600], file=sys.stderr)


if __name__ == "__main__":
    main()

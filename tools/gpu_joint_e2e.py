"""Wall time of the joint command from files to files on a config-3-like amplicon set
(pipeline.quantify_joint: round 1 -> grid rounds 2/3 -> repeat_size.txt -> 2D GMM phasing)."""
import json, os, sys, tempfile, time
import numpy as np
sys.path.insert(0, '.')
from nanorepeat_amd import synth, pipeline, joint as J, io as IO, phasing

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
j = synth.config3(n)
left, u1, mid, u2, right = j["region"]
chrom = left + u1 * 19 + mid + u2 * 7 + right
tmp = tempfile.mkdtemp(prefix="nra_joint_")
open(os.path.join(tmp, "ref.fa"), "w").write(">chr4\n" + chrom + "\n")
with open(os.path.join(tmp, "reads.fastq"), "w") as f:
    for i, s in enumerate(j["reads"]):
        f.write(f"@r{i}\n{s}\n+\n{'I' * len(s)}\n")
r1 = f"chr4:{len(left)}:{len(left) + 57}:{u1}:200"
r2 = f"chr4:{len(left) + 57 + len(mid)}:{len(left) + 57 + len(mid) + 21}:{u2}:20"
t = {}
t0 = time.time(); fq = IO.fastq_file_to_dict(os.path.join(tmp, "reads.fastq")); t["read_fastq"] = time.time() - t0
a = J.Repeat().init_from_string(r1); b = J.Repeat().init_from_string(r2); a.max_size += 10; b.max_size += 10
t0 = time.time(); init = J.initial_estimate_repeat_size(chrom, fq, "ont", 1, a, b, 1000); t["round1"] = time.time() - t0
t0 = time.time(); fin = J.fine_tune_read_count(init, fq, chrom, a, b, "ont"); t["rounds_2_3"] = time.time() - t0
t0 = time.time(); counts, _ = J.output_repeat_size_2d("reads.fastq", a.repeat_id, b.repeat_id, os.path.join(tmp, "out"), fin.repeat1_count_dict, fin.repeat2_count_dict); t["repeat_size_txt"] = time.time() - t0
t0 = time.time()
fitted = pipeline._fit_in_worker_processes([("2d", (counts, 2, 0.1, 0.1, 22, False, 1))], 1)[0]      # what quantify_joint does
alleles = phasing.split_alleles_using_gmm_2d(2, 0.1, 0.1, False, 22, a, b, counts, 0, os.path.join(tmp, "reads.fastq"), os.path.join(tmp, "out"), fitted=fitted)
t["gmm_phasing"] = time.time() - t0
k1 = np.array([fin.repeat1_count_dict.get(f"r{i}", -1) for i in range(n)])
print(json.dumps({"reads": n, "reads_with_round1_ranges": len(init.repeat1_count_range_dict), "seconds": {k: round(v, 3) for k, v in t.items()},
                  "k1_within1": float(np.mean(np.abs(k1 - j["truth"][:, 0]) <= 1)),
                  "alleles": [(x.repeat1_median_size, x.repeat2_median_size, x.num_reads) for x in (alleles or [])]}))

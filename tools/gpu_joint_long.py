"""Joint grid rounds on long amplicon reads (rows-per-lane buckets of 32 and more: one wave per SIMD).
python3 tools/gpu_joint_long.py [n_reads] [read_len]"""
import copy, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nanorepeat_amd import joint as J, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
read_len = int(sys.argv[2]) if len(sys.argv) > 2 else 2500
j = synth.make_joint(n, read_len=read_len, read_sd=120, anchor=1500, seed=5)
init = J.Round1Estimation(); fq = {}
for i, s in enumerate(j["reads"]):
    init.repeat1_count_range_dict[f"r{i}"] = tuple(int(x) for x in j["range1"][i])
    init.repeat2_count_range_dict[f"r{i}"] = tuple(int(x) for x in j["range2"][i])
    init.read_strand_dict[f"r{i}"] = int(j["strand"][i])
    fq[f"r{i}"] = f"@r{i}\n{s}\n+\n{'!' * len(s)}\n"
left, u1, mid, u2, right = j["region"]
chrom = left + u1 * 19 + mid + u2 * 7 + right
a = J.Repeat.parse(f"chr4:{len(left)}:{len(left) + 57}:{u1}:200")
b = J.Repeat.parse(f"chr4:{len(left) + 57 + len(mid)}:{len(left) + 57 + len(mid) + 21}:{u2}:20")
a.max_size += 10; b.max_size += 10
sess = J.GridSession(J._joint_region(chrom, a, b, max_flanking_len=1500), fq, parts=1)
sess.rounds = []
ts = []
for it in range(6):
    sess.new_run()
    t0 = time.perf_counter()
    est = J.fine_tune_read_count(init, fq, chrom, copy.copy(a), copy.copy(b), session=sess)
    ts.append(1e3 * (time.perf_counter() - t0))
st = sess.rounds[-1][1]
dev = [r[1]["total_ms"] for r in sess.rounds[-2:]]
print(json.dumps({"reads": n, "read_len": read_len, "ms_per_run": ts, "device_ms_last_two_rounds": dev,
                  "mean_read_len": sum(map(len, j["reads"])) / n}))
sess.close()

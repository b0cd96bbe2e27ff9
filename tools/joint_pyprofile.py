import cProfile, pstats, copy, os, sys
sys.path.insert(0, os.getcwd())
import nanorepeat_amd
nanorepeat_amd.apply_recommended_env()      # (like bench.py: 8 hardware queues)
from nanorepeat_amd import joint as J, synth
n = 5000
j = synth.config3(n)
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
session = J.GridSession(J._joint_region(chrom, a, b), fq, parts=1)
for it in range(3):
    session.new_run(); J.fine_tune_read_count(init, fq, chrom, copy.copy(a), copy.copy(b), session=session)
pr = cProfile.Profile()
pr.enable()
for it in range(30):
    session.new_run(); J.fine_tune_read_count(init, fq, chrom, copy.copy(a), copy.copy(b), session=session)
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
session.close()

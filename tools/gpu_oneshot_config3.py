"""Where one joint scorer call (joint.fine_tune_read_count from the FASTQ dict, no resident session) spends its time:
session creation (packing, H2D, device buffers) against the two grid rounds.  NRA_DEBUG_PHASES=1 adds the library's marks.
python3 tools/gpu_oneshot_config3.py [n_reads = 5000]"""
import copy, json, sys, time
sys.path.insert(0, '.')
import nanorepeat_amd
nanorepeat_amd.apply_recommended_env()
from nanorepeat_amd import joint as J, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
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
whole, create, rounds, close = [], [], [], []
for it in range(8):
    t0 = time.perf_counter()
    J.fine_tune_read_count(init, fq, chrom, copy.copy(a), copy.copy(b))
    whole.append(round(1e3 * (time.perf_counter() - t0), 2))
    print("---- call done", file=sys.stderr, flush=True)
for it in range(6):
    t0 = time.perf_counter()
    s = J.GridSession(J._joint_region(chrom, a, b), fq)
    t1 = time.perf_counter()
    J.fine_tune_read_count(init, fq, chrom, copy.copy(a), copy.copy(b), session=s)
    t2 = time.perf_counter()
    s.close()
    t3 = time.perf_counter()
    create.append(round(1e3 * (t1 - t0), 2)); rounds.append(round(1e3 * (t2 - t1), 2)); close.append(round(1e3 * (t3 - t2), 2))
print(json.dumps({"reads": n, "whole_call_ms": whole, "session_create_ms": create, "rounds_ms": rounds, "close_ms": close}))

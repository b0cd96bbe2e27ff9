"""One-off: the HIP joint path against the CPU oracle on a slice of BASELINE config 3, both grid
rounds through the host mirror: per-cell (score, window score), per-read tie sums, strands.
Usage: python tools/gpu_validate_config3.py [n_reads]"""
import copy, json, sys, time
import numpy as np
sys.path.insert(0, '.')
from nanorepeat_amd import _capi as A, synth, joint as J
from oracle import oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
j = synth.config3(n)
init = J.Round1Estimation(); fq = {}
for i, s in enumerate(j["reads"]):
    init.repeat1_count_range_dict[f"r{i}"] = tuple(int(x) for x in j["range1"][i])
    init.repeat2_count_range_dict[f"r{i}"] = tuple(int(x) for x in j["range2"][i])
    fq[f"r{i}"] = f"@r{i}\n{s}\n+\n{'!' * len(s)}\n"
left, u1, mid, u2, right = j["region"]
chrom = left + u1 * 19 + mid + u2 * 7 + right
a = J.Repeat().init_from_string(f"chr4:{len(left)}:{len(left) + 57}:{u1}:200")
b = J.Repeat().init_from_string(f"chr4:{len(left) + 57 + len(mid)}:{len(left) + 57 + len(mid) + 21}:{u2}:20")
a.max_size += 10; b.max_size += 10
KEYS = ("read_strand", "cell_score", "cell_wscore", "best_wscore", "sum_k1", "sum_k2", "n_ties", "status")
rounds = []

def both(region, reads, cr, k1, k2, **kw):
    g = A.joint_2d(region, reads, cr, k1, k2, read_strand=kw.get("read_strand"))
    t0 = time.time()
    o = O.joint_2d(region, reads, cr, k1, k2, read_strand=kw.get("read_strand"))
    rounds.append({"cells": len(cr), "cpu_s": time.time() - t0, **{k: bool(np.array_equal(g[k], o[k])) for k in KEYS}})
    return g

fin = J.fine_tune_read_count(init, fq, chrom, copy.deepcopy(a), copy.deepcopy(b), scorer=both)
ok = all(all(v for k, v in r.items() if k in KEYS) for r in rounds)
print(json.dumps({"reads": n, "rounds": rounds, "all_equal": ok}))
sys.exit(0 if ok else 1)

"""Times BASELINE config 3 (HTT-like joint CAG+CCG, 5k amplicon reads) through the host mirror."""
import copy, json, sys, time
import numpy as np
sys.path.insert(0, '.')
from nanorepeat_amd import _capi as A, synth, joint as J

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
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
stats = []
def timed(flags):
    def scorer(region, reads, cr, k1, k2, **kw):
        t0 = time.time()
        with A.Batch.create_2d(region, reads, cr, k1, k2, read_strand=kw.get("read_strand"), flags=flags) as bt:
            bt.run(); bt.sync()
            st = bt.stats(); out = bt.fetch()
        stats.append(dict(cells=len(cr), wall_s=time.time() - t0, device_ms=st["total_ms"], phase_ms=st["score_phase_ms"],
                          exe_cells=st["executed_cells"], alg_cells=st["algorithmic_cells"]))
        return out
    return scorer
for name, flags in (("decomposition", 0), ("brute", A.F_BRUTE_FORCE)):
    if name == "brute" and n > 1000: continue
    stats.clear()
    t0 = time.time()
    fin = J.fine_tune_read_count(init, fq, chrom, copy.deepcopy(a), copy.deepcopy(b), scorer=timed(flags))
    wall = time.time() - t0
    k1 = np.array([fin.repeat1_count_dict.get(f"r{i}", -1) for i in range(n)]); k2 = np.array([fin.repeat2_count_dict.get(f"r{i}", -1) for i in range(n)])
    print(json.dumps(dict(mode=name, reads=n, wall_s=wall, rounds=stats,
                          k1_within1=float(np.mean(np.abs(k1 - j["truth"][:, 0]) <= 1)), k2_within1=float(np.mean(np.abs(k2 - j["truth"][:, 1]) <= 1)))), flush=True)

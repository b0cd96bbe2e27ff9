"""One-off: the HIP path against the CPU oracle on a large slice of BASELINE config 2 (per-candidate
scores, tie sums, statuses and, with NRA_F_TIE_EXTENTS, the extents of every tied candidate).
Usage: python tools/gpu_validate_config2.py [n_reads]"""
import json, sys, time
import numpy as np
sys.path.insert(0, '.')
from nanorepeat_amd import _capi as A, synth
from oracle import oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
d = synth.config2()
reads, kmin, kmax = d["reads"][:n], d["kmin"][:n], d["kmax"][:n]
t0 = time.time(); o = O.round3_1d(d["regions"], reads, kmin, kmax); t_cpu = time.time() - t0
res = {"reads": n, "alignments": int((kmax.astype(np.int64) - kmin + 1).sum()), "cpu_s": t_cpu}
for name, flags in (("default", 0), ("tie_extents", A.F_TIE_EXTENTS)):
    g = A.round3_1d(d["regions"], reads, kmin, kmax, flags=flags)
    keys = ["best_score", "sum_k", "n_ties", "status", "cand_score"]
    if flags:
        tied = np.repeat(o["best_score"], kmax - kmin + 1) == o["cand_score"]
        res[name + "_tied_extents_equal"] = bool(np.array_equal(g["cand_tstart"][tied], o["cand_tstart"][tied]) and
                                                 np.array_equal(g["cand_tend"][tied], o["cand_tend"][tied]))
    res[name] = {k: bool(np.array_equal(g[k], o[k])) for k in keys}
print(json.dumps(res))
sys.exit(0 if all(all(v.values()) if isinstance(v, dict) else True for v in res.values()) and res["tie_extents_tied_extents_equal"] else 1)

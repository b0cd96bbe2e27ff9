"""One nra_round3_1d call from host buffers to host results on small batches of BASELINE config 2's region (what one of
the reference's per-region workers would hand over): the floor is launch and copy latency, not arithmetic.
Usage: python tools/gpu_small_calls.py"""
import json, sys, time
import numpy as np
sys.path.insert(0, '.')
import nanorepeat_amd
nanorepeat_amd.apply_recommended_env()
from nanorepeat_amd import _capi as A, synth

out = {}
for n in (10, 50, 100, 500, 1000, 2500, 10000):
    d = synth.config2(n_reads=n)
    n_align = int((d["kmax"].astype(np.int64) - d["kmin"] + 1).sum())
    call, _ = A.prepared_round3_1d(d["regions"], d["reads"], d["kmin"], d["kmax"])      # per-read results only, prebuilt buffers
    call()
    ts = []
    for _ in range(15):
        t0 = time.perf_counter()
        call()
        ts.append(time.perf_counter() - t0)
    med = float(np.median(ts))
    out[str(n)] = {"reads": n, "alignments": n_align, "ms_per_call": round(med * 1e3, 3), "Malign_per_s": round(n_align / med / 1e6, 2)}
    print(n, out[str(n)], flush=True)
print(json.dumps({"what": "one nra_round3_1d call (host buffers -> host results), config 2's region with n reads, median of 15", "runs": out}))

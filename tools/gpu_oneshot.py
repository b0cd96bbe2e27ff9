"""Host-buffer-inclusive rate of the 1D path: one `nra_round3_1d` call from host arrays to host
results (encode + 2-bit packing + hipMalloc + H2D + kernels + D2H + free), BASELINE config 2."""
import json, sys, time
import numpy as np
sys.path.insert(0, '.')
from nanorepeat_amd import _capi as A, synth

d = synth.config2()
n_align = int((d["kmax"].astype(np.int64) - d["kmin"] + 1).sum())
seqs, off = A.pack_reads(d["reads"])
A.round3_1d(d["regions"], d["reads"][:64], d["kmin"][:64], d["kmax"][:64])          # context, code objects
rows = []
for rep in range(5):
    t0 = time.perf_counter()
    out = A.round3_1d(d["regions"], d["reads"], d["kmin"], d["kmax"], per_candidate=False)   # what round3.py calls
    rows.append(time.perf_counter() - t0)
t_call = min(rows)
# the same through the resident-batch entry points, phase by phase
t0 = time.perf_counter(); b = A.Batch.create_1d(d["regions"], d["reads"], d["kmin"], d["kmax"]); t_create = time.perf_counter() - t0
t0 = time.perf_counter(); b.run(); b.sync(); t_run = time.perf_counter() - t0
t0 = time.perf_counter(); b.fetch(per_candidate=False); t_fetch = time.perf_counter() - t0
t0 = time.perf_counter(); b.fetch(); t_fetch_all = time.perf_counter() - t0
b.destroy() if hasattr(b, "destroy") else None
print(json.dumps({"workload": "config 2, 10k reads x 196 candidates", "alignments": n_align,
                  "one_shot_call_ms": t_call * 1e3, "one_shot_Malign_per_s": n_align / t_call / 1e6,
                  "create_ms": t_create * 1e3, "run_ms": t_run * 1e3, "fetch_per_read_ms": t_fetch * 1e3,
                  "fetch_with_per_candidate_arrays_ms": t_fetch_all * 1e3,
                  "input_bytes": int(len(seqs) + off.nbytes + d["kmin"].nbytes * 2)}))

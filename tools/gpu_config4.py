"""BASELINE config 4 at a fifth of its size (200 regions x 1000 reads) and more: scaling and memory."""
import json, sys, time
import numpy as np
sys.path.insert(0, '.')
from nanorepeat_amd import _capi as A, synth
nreg = int(sys.argv[1]) if len(sys.argv) > 1 else 200
t0 = time.time(); d = synth.config4(nreg, 1000); tg = time.time() - t0
t0 = time.time()
with A.Batch.create_1d(d["regions"], d["reads"], d["kmin"], d["kmax"], read_region=d["read_region"]) as b:
    tc = time.time() - t0
    b.run(); b.sync()
    t0 = time.time()
    for _ in range(3):
        b.run(); b.sync()
    dt = (time.time() - t0) / 3
    st = b.stats(); out = b.fetch(per_candidate=False)
ok = out["status"] == 0
est = out["sum_k"][ok] / np.maximum(out["n_ties"][ok], 1)
print(json.dumps(dict(regions=nreg, reads=len(d["reads"]), alignments=st["n_alignments"], gen_s=tg, create_s=tc, ms_per_step=dt * 1e3,
                      Malign_per_s=st["n_alignments"] / dt / 1e6, phase_ms=st["score_phase_ms"], exe_Tcells_s=st["executed_cells"] / st["score_phase_ms"] / 1e9,
                      ok=float(ok.mean()), within1=float(np.mean(np.abs(est - d["k_true"][ok]) <= 1)), extent_tasks=st["n_extent_tasks"])), flush=True)

"""Times the other BASELINE configs (3, 4 reduced, 5) through the C ABI; prints one JSON line each."""
import json, sys, time
import numpy as np
sys.path.insert(0, '.')
from nanorepeat_amd import _capi as A, synth, joint as J

def run1d(name, d, flags=0, reps=3):
    t0 = time.time()
    with A.Batch.create_1d(d["regions"], d["reads"], d["kmin"], d["kmax"], read_region=d.get("read_region"), flags=flags) as b:
        t_create = time.time() - t0
        b.run(); b.sync()
        t0 = time.time()
        for _ in range(reps):
            b.run(); b.sync()
        dt = (time.time() - t0) / reps
        st = b.stats(); out = b.fetch(per_candidate=False)
    ok = out["status"] == 0
    est = out["sum_k"][ok] / np.maximum(out["n_ties"][ok], 1)
    print(json.dumps(dict(config=name, reads=len(d["reads"]), alignments=st["n_alignments"], ms=dt * 1e3,
                          Malign_per_s=st["n_alignments"] / dt / 1e6, create_s=t_create,
                          exe_Tcells_s=st["executed_cells"] / (st["score_kernel_ms"] / 1e3) / 1e12,
                          alg_over_exe=st["algorithmic_cells"] / st["executed_cells"],
                          ok_frac=float(ok.mean()), within1=float(np.mean(np.abs(est - d["k_true"][ok]) <= 1)),
                          extent_tasks=st["n_extent_tasks"], kernel_ms=st["score_kernel_ms"], ext_ms=st["extent_kernel_ms"])), flush=True)

which = sys.argv[1:] or ["5", "4", "3"]
if "5" in which:
    run1d("config5 (hifi, k in [5,500], 1000 reads)", synth.config5(1000))
if "4" in which:
    t0 = time.time(); d = synth.config4(100, 1000); print("gen", time.time() - t0, flush=True)
    run1d("config4/10 (100 regions x 1000 reads, reference window rule)", d)
if "3" in which:
    j = synth.config3(1000)
    cr, k1, k2 = [], [], []
    for r in range(len(j["reads"])):
        for a in range(int(j["range1"][r][0]), int(j["range1"][r][1]), 4):
            for b in range(int(j["range2"][r][0]), int(j["range2"][r][1]), 4):
                cr.append(r); k1.append(a); k2.append(b)
    t0 = time.time()
    with A.Batch.create_2d(j["region"], j["reads"], cr, k1, k2) as b:
        b.run(); b.sync()
        t0 = time.time(); b.run(); b.sync(); dt = time.time() - t0
        st = b.stats(); out = b.fetch(per_candidate=False)
    print(json.dumps(dict(config="config3/5 round-2-like grid (1000 reads, step 4)", cells=len(cr), ms=dt * 1e3,
                          Mcells_per_s=len(cr) / dt / 1e6, exe_Tcells_s=st["executed_cells"] / (st["score_kernel_ms"] / 1e3) / 1e12,
                          strand_ok=float(np.mean(out["read_strand"] == j["strand"])))), flush=True)

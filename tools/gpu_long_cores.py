"""Times the 1D path on long cores (large expansions): 3.7 and 5.7 kb cores (packed chained blocks) and 7.7 kb cores
(int32 chained blocks): row blocks as concurrent waves (k_sweep_ringmt, the default) vs one wave per read pair
(NRA_F_SERIAL_CHAIN, k_sweep_ringchain) vs the DPP chain (NRA_F_DPP_SWEEP).  Usage: python tools/gpu_long_cores.py [n_reads] [modes]"""
import json, sys, time
import numpy as np
sys.path.insert(0, '.')
from nanorepeat_amd import _capi as A, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
out = {}
for name, alleles in (("cores_3.7kb", (700, 705)), ("cores_5.7kb", (1100, 1105)), ("cores_7.7kb", (1500, 1505)),
                      ("cores_2.6kb_unchained", (480, 485))):
    d = synth.make_1d(n, "TATTG", alleles, "ont_q20", kwin=None, seed=77)
    n_align = int((d["kmax"].astype(np.int64) - d["kmin"] + 1).sum())
    row = {"reads": n, "alignments": n_align}
    ref = None
    modes = (sys.argv[2].split(",") if len(sys.argv) > 2 else ["concurrent_blocks", "serial_blocks"])
    for mode, flags in (("concurrent_blocks", 0), ("serial_blocks", A.F_SERIAL_CHAIN), ("dpp", A.F_DPP_SWEEP)):
        if mode not in modes:
            continue
        with A.Batch.create_1d(d["regions"], d["reads"], d["kmin"], d["kmax"], flags=flags) as b:
            b.run(); b.sync()
            t0 = time.perf_counter()
            for _ in range(3):
                b.run()
            b.sync()
            dt = (time.perf_counter() - t0) / 3
            st = b.stats(); g = b.fetch(per_candidate=False)
        if ref is None:
            ref = g
        row[mode] = {"ms_per_pass": dt * 1e3, "Malign_per_s": n_align / dt / 1e6, "executed_Tcell_per_s": st["executed_cells"] / dt / 1e12,
                     "executed_cells": st["executed_cells"], "same_as_first": all(np.array_equal(g[k], ref[k]) for k in g)}
    out[name] = row
print(json.dumps(out, indent=1))

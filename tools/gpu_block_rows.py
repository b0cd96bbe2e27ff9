"""Reads of 1.1 - 3.0 kb in the 1D path: one register block per read (64 x R rows, R = 18 .. 48: two waves, then one wave
per SIMD) against row blocks as concurrent waves (k_sweep_ringmt, 64 x 12 .. 15 rows, three waves per SIMD), which
NRA_CHAIN_FROM=<rows> switches on from that read length.  Usage: python tools/gpu_block_rows.py [n_reads]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, '.')
import nanorepeat_amd
nanorepeat_amd.apply_recommended_env()
from nanorepeat_amd import _capi as A, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
out = {}
ks = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else (180, 240, 280, 320, 360, 400, 440, 480, 520, 560)
for k in ks:
    d = synth.make_1d(n, "TATTG", (k, k + 4), "hifi", kwin=None, seed=5)
    q = np.array([len(r) for r in d["reads"]])
    n_align = int((d["kmax"].astype(np.int64) - d["kmin"] + 1).sum())
    row = {"read_len_mean": float(q.mean()), "read_len_max": int(q.max()), "alignments": n_align}
    ref = None
    for mode, env in (("one_block", "3072"), ("row_blocks", "64"), ("library_choice", None)):
        if env is None:
            os.environ.pop("NRA_CHAIN_FROM", None)
        else:
            os.environ["NRA_CHAIN_FROM"] = env
        with A.Batch.create_1d(d["regions"], d["reads"], d["kmin"], d["kmax"]) as b:
            b.run(); b.sync()
            t0 = time.perf_counter()
            for _ in range(3):
                b.run()
            b.sync()
            dt = (time.perf_counter() - t0) / 3
            st = b.stats(); g = b.fetch(per_candidate=False)
        if ref is None:
            ref = g
        row[mode] = {"ms_per_pass": round(dt * 1e3, 3), "executed_cells": st["executed_cells"], "executed_Tcell_per_s": round(st["executed_cells"] / dt / 1e12, 3),
                     "same_results": all(np.array_equal(g[x], ref[x]) for x in g)}
    row["row_blocks_over_one_block"] = round(row["row_blocks"]["ms_per_pass"] / row["one_block"]["ms_per_pass"], 3)
    out[f"k{k}"] = row
    lib = row["library_choice"]
    took = "row blocks" if lib["executed_cells"] == row["row_blocks"]["executed_cells"] != row["one_block"]["executed_cells"] else \
           "one block" if lib["executed_cells"] == row["one_block"]["executed_cells"] != row["row_blocks"]["executed_cells"] else "either (same cells)"
    print(f"n{n} k{k} q{row['read_len_max']}: one block {row['one_block']['ms_per_pass']} ms ({row['one_block']['executed_Tcell_per_s']} T/s), "
          f"row blocks {row['row_blocks']['ms_per_pass']} ms ({row['row_blocks']['executed_Tcell_per_s']} T/s), ratio {row['row_blocks_over_one_block']}; "
          f"the library takes {took}: {lib['ms_per_pass']} ms", flush=True)

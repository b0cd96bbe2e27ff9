"""One-off: a joint batch whose kept column states exceed 32 GiB (default budget: kept) against the same batch with
NRA_F_JOINT_NO_KEEP, both grid rounds -- every per-read result and every cell equal, fewer cells executed in round 3.
python3 tools/gpu_big_batch_check.py [n_reads = 70000]"""
import json, sys, time
import numpy as np
sys.path.insert(0, '.')
from nanorepeat_amd import _capi as capi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 70000
j = synth.make_joint(n, seed=5)
t1, t2 = j["truth"][:, 0].astype(np.float64), j["truth"][:, 1].astype(np.float64)
strands = j["strand"].astype(np.int8)
coarse = capi.Grid((0, 4, 25), t1 - 20, t1 + 13, (0, 3, 8), np.zeros(len(t2)), t2 + 8)
fine = capi.Grid((0, 1, 90), t1 - 2, t1 + 3, (0, 1, 24), np.maximum(t2 - 2, 0), t2 + 2)
got, ms = {}, {}
for name, flags in (("default", 0), ("no keep", capi.F_JOINT_NO_KEEP)):
    with capi.Batch.create_2d_reads(j["region"], j["reads"], flags=flags) as b:
        for grid in (coarse, fine):
            t0 = time.perf_counter()
            assert b.set_grid(grid, strands) > 0
            b.run(); b.sync()
            ms.setdefault(name, []).append(round(1e3 * (time.perf_counter() - t0), 1))
            got.setdefault(name, []).append((b.fetch(per_candidate=True), b.stats()))
same = all(np.array_equal(a[key], c[key]) for (a, _), (c, _) in zip(got["default"], got["no keep"]) for key in a)
cells = {k: [int(st["executed_cells"]) for _, st in v] for k, v in got.items()}
kept_gib = got["default"][0][1].get("intermediate_bytes", 0) / 2 ** 30
ok = same and cells["default"][1] < 0.5 * cells["no keep"][1]
print(json.dumps({"reads": n, "equal_results_and_cells": bool(same), "executed_cells": cells, "ms_per_grid_first_call": ms,
                  "intermediate_GiB_round2": round(kept_gib, 1), "ok": bool(ok)}))
sys.exit(0 if ok else 1)

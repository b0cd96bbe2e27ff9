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

def score_grid(self, grid, read_strand):
    """The product path (routed grid on the resident batch: junction at the end of mid, scans) with the per-cell
    arrays fetched too, against the oracle on the cell list the library's routing gives."""
    n_cells = self.batch.set_grid(grid, read_strand)
    self.batch.run(); self.batch.sync()
    g = self.batch.fetch(per_candidate=True)
    cr, k1, k2 = A.joint_grid_cells(grid)
    t0 = time.time()
    o = O.joint_2d(self.region, self.reads, cr, k1, k2, read_strand=read_strand)
    has = np.zeros(len(self.reads), bool); has[cr] = True
    rounds.append({"cells": int(n_cells), "cpu_s": time.time() - t0, "executed_cells": int(self.batch.stats()["executed_cells"]),
                   **{k: bool(np.array_equal(np.asarray(g[k])[has if len(o[k]) == len(has) else slice(None)],
                                             np.asarray(o[k])[has if len(o[k]) == len(has) else slice(None)])) for k in KEYS}})
    return g, n_cells

J.GridSession.score_grid = score_grid
for i in range(n):
    init.read_strand_dict[f"r{i}"] = int(j["strand"][i])
fin = J.fine_tune_read_count(init, fq, chrom, copy.deepcopy(a), copy.deepcopy(b))
ok = all(all(v for k, v in r.items() if k in KEYS) for r in rounds)
print(json.dumps({"reads": n, "path": "nra_batch2d_set_grid (junction at the end of mid, k_joint_midscan, k_joint_combine; strands given: round 3 from the column states round 2 kept, no sweeps -- see executed_cells)", "rounds": rounds, "all_equal": ok}))
sys.exit(0 if ok else 1)

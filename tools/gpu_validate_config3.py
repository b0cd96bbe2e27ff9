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

def score_grid(self, grid, read_strand, refine=None):
    """The product path (routed grid on the resident batch: junction at the end of mid, scans) with the per-cell
    arrays fetched too, against the oracle on the cell list the library's routing gives.  refine given: the reference's
    round 3 enqueued behind the grid and routed on the device (nra_batch2d_refine) -- the oracle then scores the finer grid
    the host would have routed from the ORACLE's round-2 results."""
    n_cells = self.batch.set_grid(grid, read_strand)
    self.batch.run()
    cr, k1, k2 = A.joint_grid_cells(grid)
    t0 = time.time()
    o = O.joint_2d(self.region, self.reads, cr, k1, k2, read_strand=read_strand)
    self.refined = refine is not None and self.batch.refine(*refine)
    keys = KEYS
    if self.refined:
        s1, s2, lo1, hi1, lo2, hi2 = refine
        ok = (o["status"] == 0) & (o["n_ties"] > 0)
        nt = np.maximum(o["n_ties"], 1).astype(np.float64)
        z1, z2 = o["sum_k1"] / nt, o["sum_k2"] / nt
        f = [np.where(ok, v, 0.0) for v in (np.maximum(z1 - s1, lo1), np.minimum(z1 + s1, hi1), np.maximum(z2 - s2, lo2), np.minimum(z2 + s2, hi2))]
        cr, k1, k2 = A.joint_grid_cells(A.Grid((0, 1, 400), f[0], f[1], (0, 1, 100), f[2], f[3]))
        o = O.joint_2d(self.region, self.reads, cr, k1, k2, read_strand=read_strand)
        keys = tuple(k for k in KEYS if not k.startswith("cell_"))       # (the refinement's cells are laid out per read)
    self.batch.sync()
    g = self.batch.fetch(per_candidate=True)
    st = self.batch.stats()
    has = np.zeros(len(self.reads), bool); has[cr] = True
    rec = {"cells": int(st["n_alignments"]), "refined_on_device": bool(self.refined), "cpu_s": time.time() - t0, "executed_cells": int(st["executed_cells"]),
           **{k: bool(np.array_equal(np.asarray(g[k])[has if len(o[k]) == len(has) else slice(None)],
                                     np.asarray(o[k])[has if len(o[k]) == len(has) else slice(None)])) for k in keys}}
    if self.refined:
        cap = 4 * s1 * s2
        cs, cw = g["cell_score"].reshape(len(self.reads), cap), g["cell_wscore"].reshape(len(self.reads), cap)
        same = True
        for r in np.nonzero(has)[0]:
            mine = np.nonzero(cr == r)[0]
            same &= bool(np.array_equal(cs[r, :len(mine)], o["cell_score"][mine]) and np.array_equal(cw[r, :len(mine)], o["cell_wscore"][mine]))
        rec["cell_score"] = rec["cell_wscore"] = same
        rec["cells_refinement"] = int(len(cr))
    rounds.append(rec)
    return g, int(st["n_alignments"])

J.GridSession.score_grid = score_grid
for i in range(n):
    init.read_strand_dict[f"r{i}"] = int(j["strand"][i])
fin = J.fine_tune_read_count(init, fq, chrom, copy.deepcopy(a), copy.deepcopy(b))                      # round 3 behind round 2 on the device
two = J.fine_tune_read_count(init, fq, chrom, copy.deepcopy(a), copy.deepcopy(b), refine=False)        # two grid calls
same = dict(fin.repeat1_count_dict) == dict(two.repeat1_count_dict) and dict(fin.repeat2_count_dict) == dict(two.repeat2_count_dict)
ok = all(all(v for k, v in r.items() if k in KEYS) for r in rounds) and same and rounds[0]["refined_on_device"]
print(json.dumps({"reads": n, "path": "flank sweeps ahead of the grids; run 1: round 2 + round 3 routed on the device (nra_batch2d_refine), run 2: two nra_batch2d_set_grid calls (round 3 from the column states round 2 kept)", "rounds": rounds, "both_paths_same_sizes": same, "all_equal": ok}))
sys.exit(0 if ok else 1)

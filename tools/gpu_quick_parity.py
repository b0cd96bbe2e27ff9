import sys, time, numpy as np
sys.path.insert(0, '.')
from nanorepeat_amd import _capi as A, synth
from oracle import oracle as O
print("devices", A.device_count(), A.load().nra_version())
d = synth.make_1d(24, "TATTG", (8, 30), "ont", kwin=(0, 45), flank=100, anchor=300, seed=1)
t0=time.time(); g = A.round3_1d(d["regions"], d["reads"], d["kmin"], d["kmax"]); print("gpu", time.time()-t0)
t0=time.time(); o = O.round3_1d(d["regions"], d["reads"], d["kmin"], d["kmax"]); print("cpu", time.time()-t0)
for k in g:
    same = np.array_equal(g[k], o[k])
    print(k, same)
    if not same:
        idx = np.nonzero(g[k] != o[k])[0][:10]; print(idx, g[k][idx], o[k][idx])
print(g["sum_k"][:10], g["n_ties"][:10], d["k_true"][:10], g["status"][:10])
g2 = A.round3_1d(d["regions"], d["reads"], d["kmin"], d["kmax"], flags=1)
o2 = O.round3_1d(d["regions"], d["reads"], d["kmin"], d["kmax"], flags=1)
for k in g2: print("allext", k, np.array_equal(g2[k], o2[k]))
# 2D
j = synth.make_joint(12, alleles=((6, 4), (11, 3)), read_len=500, read_sd=30, anchor=300, seed=3)
cr, k1, k2 = [], [], []
for r in range(len(j["reads"])):
    for a in range(j["range1"][r][0], j["range1"][r][1], 2):
        for b in range(j["range2"][r][0], j["range2"][r][1], 2):
            cr.append(r); k1.append(a); k2.append(b)
gj = A.joint_2d(j["region"], j["reads"], cr, k1, k2)
oj = O.joint_2d(j["region"], j["reads"], cr, k1, k2)
for k in gj:
    same = np.array_equal(gj[k], oj[k]); print("2d", k, same)
    if not same:
        idx = np.nonzero(gj[k] != oj[k])[0][:10]; print(idx, gj[k][idx], oj[k][idx])
print(gj["read_strand"], j["strand"])
print(gj["sum_k1"]/np.maximum(gj["n_ties"],1), gj["sum_k2"]/np.maximum(gj["n_ties"],1), j["truth"].T)

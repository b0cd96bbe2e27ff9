"""Host phases of one-shot calls (NRA_DEBUG=1 prints the create phases): config 2 nra_round3_1d, config 3 joint session."""
import sys, time, copy
import numpy as np
sys.path.insert(0, '.')
from nanorepeat_amd import _capi as A, synth, joint as J

d = synth.config2()
call, out = A.prepared_round3_1d(d["regions"], d["reads"], d["kmin"], d["kmax"])
call(); call()
print("=== config 2 one-shot call", file=sys.stderr, flush=True)
t0 = time.perf_counter(); call(); print(f"call {1e3 * (time.perf_counter() - t0):.3f} ms", file=sys.stderr, flush=True)
b = A.Batch.create_1d(d["regions"], d["reads"], d["kmin"], d["kmax"])
for _ in range(2):
    t0 = time.perf_counter(); b.run(); b.sync(); t1 = time.perf_counter(); b.fetch(per_candidate=False); t2 = time.perf_counter()
    print(f"resident run+sync {1e3 * (t1 - t0):.3f} ms, fetch {1e3 * (t2 - t1):.3f} ms", file=sys.stderr, flush=True)
b.close()

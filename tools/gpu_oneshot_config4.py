"""Host phases of ONE nra_round3_1d call at BASELINE config 4's size (1 M reads, 1000 regions): run with NRA_DEBUG_PHASES=1
to see the library's marks on stderr.  python3 tools/gpu_oneshot_config4.py [regions = 1000]"""
import json, sys, time
sys.path.insert(0, '.')
import nanorepeat_amd
nanorepeat_amd.apply_recommended_env()
from nanorepeat_amd import _capi as A, synth
nreg = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
d = synth.config4(nreg, 1000)
call, res = A.prepared_round3_1d(d["regions"], d["reads"], d["kmin"], d["kmax"], read_region=d["read_region"])
times = []
for _ in range(3):
    t0 = time.perf_counter(); call(); times.append(round(1e3 * (time.perf_counter() - t0), 1))
    print("---- call done", file=sys.stderr, flush=True)
print(json.dumps({"regions": nreg, "ms_per_call": times}))

"""Host/device timeline of one joint run (config 3): when the session (or each read group's thread, parts > 1) routes a
grid, waits for the device and reads the results.  python3 tools/joint_timeline.py [parts]
Environment: NRA_TIMELINE_READS (5000), NRA_TIMELINE_RUNS (14: every run's wall time is printed -- the bistable state of
DESIGN 9(2b) shows there), NRA_TIMELINE_FLAGS (batch flags, e.g. 1024 = NRA_F_JOINT_NO_KEEP), NRA_TIMELINE_BALLAST_GB (device
memory held beside the session, never touched)."""
import copy, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nanorepeat_amd
nanorepeat_amd.apply_recommended_env()      # (like bench.py: 8 hardware queues)
from nanorepeat_amd import joint as J, synth

parts = int(sys.argv[1]) if len(sys.argv) > 1 else 1
if len(sys.argv) > 2 and sys.argv[2] == "torch":      # like bench.py: torch's HIP context first
    import torch
    torch.cuda.synchronize()
ballast = float(os.environ.get("NRA_TIMELINE_BALLAST_GB", "0"))     # device memory held but never touched by the kernels
if ballast > 0:
    import torch
    hold = [torch.empty(int(min(ballast - g, 8.0) * (1 << 30)), dtype=torch.uint8, device="cuda").zero_() for g in range(0, int(ballast + 7.999), 8) if ballast - g > 0]
    torch.cuda.synchronize()
n = int(os.environ.get("NRA_TIMELINE_READS", "5000"))
n_runs = int(os.environ.get("NRA_TIMELINE_RUNS", "14"))
j = synth.config3(n)
init = J.Round1Estimation(); fq = {}
for i, s in enumerate(j["reads"]):
    init.repeat1_count_range_dict[f"r{i}"] = tuple(int(x) for x in j["range1"][i])
    init.repeat2_count_range_dict[f"r{i}"] = tuple(int(x) for x in j["range2"][i])
    init.read_strand_dict[f"r{i}"] = int(j["strand"][i])
    fq[f"r{i}"] = f"@r{i}\n{s}\n+\n{'!' * len(s)}\n"
left, u1, mid, u2, right = j["region"]
chrom = left + u1 * 19 + mid + u2 * 7 + right
a = J.Repeat.parse(f"chr4:{len(left)}:{len(left) + 57}:{u1}:200")
b = J.Repeat.parse(f"chr4:{len(left) + 57 + len(mid)}:{len(left) + 57 + len(mid) + 21}:{u2}:20")
a.max_size += 10; b.max_size += 10
log = []
t_refine = []
def score_grid(self, grid, read_strand, refine=None):
    me = threading.current_thread().name
    t = [time.perf_counter()]
    n_cells = self.batch.set_grid(grid, read_strand); t.append(time.perf_counter())
    self.batch.run()
    tr = time.perf_counter()
    self.refined = refine is not None and self.batch.refine(*refine)
    t.append(time.perf_counter())
    t_refine.append(1e3 * (t[-1] - tr))
    lk = self.in_turn
    if lk: self.host_lock.release()
    self.batch.sync()
    if lk: self.host_lock.acquire()
    t.append(time.perf_counter())
    out = self.batch.fetch(per_candidate=False); t.append(time.perf_counter())
    log.append((me, t))
    return out, n_cells


J.GridSession.score_grid = score_grid
session = J.GridSession(J._joint_region(chrom, a, b), fq, parts=parts, flags=int(os.environ.get("NRA_TIMELINE_FLAGS", "0")))
runs = []
stamps = []
for it in range(n_runs):
    session.new_run()
    log.clear()
    t0 = time.perf_counter()
    J.fine_tune_read_count(init, fq, chrom, copy.copy(a), copy.copy(b), session=session)
    t1 = time.perf_counter()
    runs.append(round(1e3 * (t1 - t0), 1))
    stamps.append(round(time.time(), 3))
print("runs (ms):", runs)
print("run ends (unix s):", stamps)
print(f"parts {parts}: last run {1e3 * (t1 - t0):.2f} ms")
for me, t in sorted(log, key=lambda x: x[1][0]):
    print(f"{me:12s} set_grid {1e3 * (t[0] - t0):7.2f} -> {1e3 * (t[1] - t0):7.2f}  run -> {1e3 * (t[2] - t0):7.2f}  "
          f"sync -> {1e3 * (t[3] - t0):7.2f}  fetch -> {1e3 * (t[4] - t0):7.2f}")
print("refine call (ms):", [round(x, 2) for x in t_refine[-4:]])
session.close()

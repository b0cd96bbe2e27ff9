#!/usr/bin/env python3
"""bench.py -- headline benchmark of the repeat-size scoring path on MI355X.

Metric (BASELINE.json): read-alignments/sec = (reads x candidate-k) scored per second.

--config 2 (default; BASELINE.json configs[1]): 10 000 synthetic ONT-error core reads over one 5 bp
    motif (TATTG), every read scored against k in [5,200] (196 candidates), alleles k=40/150.
    N > 1: every rank owns one such region of 10 000 reads (weak scaling).
--config 4 (BASELINE.json configs[3]): 1000 regions x 1000 reads, mixed 3-6 bp motifs, reference
    window rule; the regions are dealt to the ranks by executed DP cells (strong scaling).
--config 3 (BASELINE.json configs[2]): joint CAG+CCG grid rounds 2+3 on 5000 amplicon reads, N = 1.
--config 5 (BASELINE.json configs[4]): HiFi error model, k in [5,500] wide sweep (496 candidates), cores of
    0.5 / 2.3 kb; like config 2 one region of --reads reads per rank.

A "step" is one pass of the whole hot path over the rank's shard, whose inputs are already
resident in HBM (nanorepeat_amd.dist.ShardedBatch1D = nra_batch1d_create has run): every scoring
kernel, the per-read selection (best score, flank test, tie mean) on the device, the fetch of the
per-read results to the host and -- N > 1 -- the one all_gather that leaves them on every rank.
The same product class runs at every N; there is no data-path collective.

ONE JSON line (< 8 KB), flat where the driver keeps scalars:
  value / ms_per_step                      the resident-input rate the benchmark contract asks for
  value_scorer_call / ms_per_scorer_call   SURVEY.md 8(d): ONE scorer call from host buffers to host results (median; N = 1)
  roofline.frac                            COUNTED: SQ_INSTS_VALU of the scoring kernels x 64 lanes / scoring-phase time / peak;
                                           frac_priced (executed cells x 10 lane-ops), frac_useful (unpadded cells x 7.25 / 15.5),
                                           traffic (FETCH_SIZE + WRITE_SIZE), frac_over_issue_ceiling -- DESIGN.md 5
  cpu_baseline                             the oracle on a bounded sample of the same workload, the box's host cores
  configs.config3 / 4 / 5                  the default run (N = 1, config 2): the same record for the other BASELINE workloads,
                                           each with its own rocprofv3 --pmc child passes (--sub-configs none: skip them)
  multi_gpu, configs.config4               N > 1: per-rank kernel ms, imbalance of executed cells, the exposed all_gather of one
                                           pass; and BASELINE config 4 (1 M reads x 1000 regions) dealt over the N ranks:
                                           the strong-scaling workload north_star names

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|3|4|5] [--sub-configs 3,4,5|none]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import copy
import hashlib
import json
import os
import subprocess
import sys
import time

# before torch starts the HIP runtime: nanorepeat_amd.RECOMMENDED_ENV (an entry point's choice, not the library's)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Integer-VALU roofline of the dominant kernels:
#   peak lane-ops/s = 256 CU x 4 SIMD x 32 lanes/clk x 2.4 GHz = 78.6 T
#   roofline.frac          COUNTED: VALU wave instructions the scoring kernels issue (SQ_INSTS_VALU of rocprofv3 --pmc child
#                          runs of this very command, or of the tracked summary of the same kernel sources) x 64 lanes /
#                          the scoring phase's HIP-event time / peak
#   roofline.frac_priced   cells the kernels EXECUTE (row padding, pipeline fill included) x 10 lane-ops (SURVEY 8d's price)
#   roofline.frac_useful   cells of the decomposition WITHOUT padding and fill x what the recurrence costs: 7.25 lane-ops
#                          in the packed int16 cells (14.5 instructions per cell pair), 15.5 in the int32 cells
VALU_PEAK_TLANEOPS = 256 * 4 * 32 * 2.4e9 / 1e12      # 78.6
LANEOPS_PER_CELL = 10.0
LANEOPS_PACKED_CELL, LANEOPS_INT32_CELL = 7.25, 15.5
HBM_PEAK_GBPS = 8000.0
PROFILE_ROUND = "r04"


def pmc_profile_path(config):
    """The tracked PMC summary of `bench.py --config N` (tools/profile_bench.sh -> tools/pmc_summary.py)."""
    tag = PROFILE_ROUND if config == 2 else f"{PROFILE_ROUND}_config{config}"
    return os.path.join("profiles", f"{tag}_pmc_traffic.json")


def kernel_source_sha16():
    """Identity of the kernel sources a PMC summary belongs to: counters are only quoted when the summary was
    taken from the sources this run was built from (tools/pmc_summary.py stores the same hash)."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "nanorepeat_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".cpp", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(csrc, name), "rb").read())
    return h.hexdigest()[:16]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=2, choices=(2, 3, 4, 5))
    ap.add_argument("--reads", type=int, default=10000, help="config 2: reads per GPU (10000); config 3: reads (5000)")
    ap.add_argument("--regions", type=int, default=1000, help="config 4: regions in the whole job")
    ap.add_argument("--reads-per-region", type=int, default=1000, help="config 4")
    ap.add_argument("--brute", action="store_true",
                    help="score K independent alignments per read (k_score_pk16) instead of the decomposition")
    ap.add_argument("--no-quanta", action="store_true",
                    help="1D, comparison: reverse and forward sweeps as two launches per bucket (NRA_F_NO_QUANTA) instead of one launch of quanta taken by ticket")
    ap.add_argument("--joint-tails", action="store_true",
                    help="config 3, comparison: tail sweeps with the junction at R[0] (NRA_F_JOINT_TAILS) instead of the junction at the end of mid")
    ap.add_argument("--joint-no-chain", action="store_true",
                    help="config 3, comparison: one MID sweep per (read, k1) (NRA_F_JOINT_NO_CHAIN) instead of a read's MID sweeps chained in one wave")
    ap.add_argument("--joint-no-keep", action="store_true",
                    help="config 3, comparison: round 3 sweeps again (NRA_F_JOINT_NO_KEEP) instead of running from the column states round 2 kept")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL)")
    ap.add_argument("--same-device", action="store_true",
                    help="rehearsal only: every rank uses GPU 0 (with --backend gloo on a one-GPU box)")
    ap.add_argument("--cpu-sample", type=int, default=-1,
                    help="reads in the CPU-baseline sample (-1: sized for ~15 s, ~6 s in a sub-record; 0: skip)")
    ap.add_argument("--one-shot-calls", type=int, default=5, help="host-buffers-in/out scorer calls timed at N = 1 (0: skip)")
    ap.add_argument("--live-pmc", default="auto", choices=("auto", "on", "off"),
                    help="count the VALU instructions of this very build with a short rocprofv3 --pmc child run (auto: the "
                         "default N = 1 run only; falls back to the tracked summary when rocprofv3 is not there)")
    ap.add_argument("--sub-configs", default="3,4,5",
                    help="default run (N = 1, config 2) only: the other configs measured into `configs` (none: skip)")
    return ap.parse_args()


def spawn_ranks_if_needed(args):
    """`python bench.py --gpus N` without a launcher: start N ranks (before anything touches the
    GPU) and leave with their exit code.  A launcher's WORLD_SIZE must agree with --gpus."""
    ws = os.environ.get("WORLD_SIZE")
    if ws is not None:
        if int(ws) != args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={ws}")
        return
    if args.gpus <= 1:
        return
    port = 29000 + os.getpid() % 3000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd))


def host_cores():
    """Threads for the CPU baseline: the process's CPU share (affinity, cgroup quota), capped at
    the 16 cores a one-GPU box grants; NRA_CPU_THREADS overrides."""
    if os.environ.get("NRA_CPU_THREADS"):
        return max(1, int(os.environ["NRA_CPU_THREADS"]))
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return min(n, 16)


def cpu_baseline_1d(data, n_sample, seconds):
    """Times the CPU oracle (oracle/, the restatement of the reference algorithm: K independent
    optimal alignments per read) on a bounded sample of the same workload, all host cores."""
    import numpy as np
    from oracle import oracle as O
    cores = host_cores()
    n_total = len(data["reads"])
    rr = data.get("read_region")
    sub = lambda n: dict(read_region=None if rr is None else rr[:n])
    n_cal = min(cores, n_total)
    t0 = time.perf_counter()
    O.round3_1d(data["regions"], data["reads"][:n_cal], data["kmin"][:n_cal], data["kmax"][:n_cal], threads=cores, **sub(n_cal))
    t_cal = max(time.perf_counter() - t0, 1e-3)
    if n_sample < 0:
        n_sample = int(n_cal * seconds / t_cal)
        n_sample = max(cores, n_sample // cores * cores)
    n_sample = min(n_sample, n_total)
    if n_sample == 0:
        return None
    reads = data["reads"][:n_sample]
    kmin, kmax = data["kmin"][:n_sample], data["kmax"][:n_sample]
    t0 = time.perf_counter()
    out = O.round3_1d(data["regions"], reads, kmin, kmax, threads=cores, **sub(n_sample))
    dt = time.perf_counter() - t0
    n_align = int((kmax.astype("int64") - kmin + 1).sum())
    rec = {"value": n_align / dt, "unit": "read-alignments/s", "cores": cores, "kind": "port",
           "sample": f"first {n_sample} reads = {n_align} alignments in {dt:.1f} s; oracle/nr_oracle.c (optimal DP, K "
                     f"independent alignments per read, OpenMP), not minimap2"}
    # second leg: the SAME algorithm as the HIP sweeps (junction decomposition, oracle/nr_decomp.c, scalar C + OpenMP) on a
    # larger sample -- separates what the decomposition buys from what the hardware buys
    try:
        if all(len(g[0]) >= 1 and len(g[2]) >= 1 for g in data["regions"]):
            m = min(n_total, max(n_sample, 64 * cores))
            t0 = time.perf_counter()
            dec = O.round3_1d_decomposed(data["regions"], data["reads"][:m], data["kmin"][:m], data["kmax"][:m], threads=cores, **sub(m))
            t1 = time.perf_counter() - t0
            m2 = min(n_total, max(m, int(m * (seconds / 3.0) / max(t1, 1e-3)) // cores * cores))      # ~ a third of the oracle's time
            if m2 > m:
                t0 = time.perf_counter()
                dec = O.round3_1d_decomposed(data["regions"], data["reads"][:m2], data["kmin"][:m2], data["kmax"][:m2], threads=cores, **sub(m2))
                t1, m = time.perf_counter() - t0, m2
            na = int(np.maximum(data["kmax"][:m].astype("int64") - data["kmin"][:m] + 1, 0).sum())
            rec.update(same_algorithm_value=na / t1, same_algorithm_reads=m,
                       same_algorithm_matches_oracle=bool(all(np.array_equal(dec[k][:n_sample], out[k]) for k in ("best_score", "sum_k", "n_ties", "status"))),
                       same_algorithm_what="oracle/nr_decomp.c: the junction decomposition as scalar C, OpenMP")
            rec["_decomposed"] = (m, dec)
    except Exception as e:
        rec["same_algorithm_error"] = f"{type(e).__name__}: {e}"
    return rec, out


def live_pmc(config, kernel_substr, steps=2, timeout_s=180, extra=()):
    """Counters of the scoring kernels of THIS build, counted by short child runs `rocprofv3 --pmc <group> --kernel-trace --
    python3 bench.py --config N --steps S` (counters only, one group per pass, as MI355X_MICROARCH.md prescribes):
    SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU / GRBM_GUI_ACTIVE, then FETCH_SIZE, then WRITE_SIZE.  Returns the keys of a
    tracked summary (tools/pmc_summary.py), or None when the first pass fails (no rocprofv3, a refusal, a timeout)."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None
    tmp = os.environ.get("TMPDIR", "/tmp")

    def one_pass(counters):
        out = tempfile.mkdtemp(prefix="nra_pmc_", dir=tmp)
        cmd = [exe, "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv", "-d", out, "--", sys.executable,
               os.path.abspath(__file__), "--config", str(config), "--steps", str(steps), "--warmup", "0", "--cpu-sample", "0",
               "--one-shot-calls", "0", "--sub-configs", "none", "--live-pmc", "off"] + list(extra)
        try:
            env = dict(os.environ, TMPDIR=tmp)
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
                env.pop(k, None)
            r = subprocess.run(cmd, cwd=tmp, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s,
                               start_new_session=True)
            if r.returncode != 0:
                return None
            total, ns = {}, 0.0
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if kernel_substr not in row["Kernel_Name"]:
                        continue
                    total[row["Counter_Name"]] = total.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                    if row["Counter_Name"] == counters[-1]:
                        ns += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
            total["_ns"] = ns
            return total
        except Exception:
            return None
        finally:
            shutil.rmtree(out, ignore_errors=True)

    sq = one_pass(["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE"])
    if not sq or not sq.get("SQ_INSTS_VALU") or sq["_ns"] <= 0:
        return None
    res = {"source_sha16": kernel_source_sha16(), "live": True,
           "sweep_kernels": {"valu_wave_instructions_per_step": sq["SQ_INSTS_VALU"] / steps},
           "simd_cycles_per_valu_instruction_active": 4.0 * sq.get("SQ_ACTIVE_INST_VALU", 0.0) / sq["SQ_INSTS_VALU"],
           "clock_GHz": sq["GRBM_GUI_ACTIVE"] / 8.0 / sq["_ns"]}
    fe, wr = one_pass(["FETCH_SIZE"]), one_pass(["WRITE_SIZE"])
    if fe and wr and "FETCH_SIZE" in fe and "WRITE_SIZE" in wr:        # KB units (MI355X_MICROARCH.md, HBM)
        res["sweep_kernels"]["fetch_bytes"] = fe["FETCH_SIZE"] * 1024 / steps
        res["sweep_kernels"]["write_bytes"] = wr["WRITE_SIZE"] * 1024 / steps
        res["hbm_bytes_per_step_sweep_kernels"] = (fe["FETCH_SIZE"] + wr["WRITE_SIZE"]) * 1024 / steps
    return res


LIVE_PMC = {}        # config -> counters of a live child run (filled by main() before the timed run)
PMC_KERNELS = {2: "k_sweep_", 3: "k_joint", 4: "k_sweep_", 5: "k_sweep_"}
PMC_STEPS = {2: 2, 3: 2, 4: 1, 5: 2}


def pmc_counters(config, brute):
    """The counters of this command: counted in this run (LIVE_PMC), else the tracked rocprofv3 --pmc summary -- quoted only
    when it was taken from the very kernel sources this run was built from.  -> (dict or None, where it came from)."""
    rel = pmc_profile_path(config)
    if brute:
        return None, "no PMC summary for this command"
    live = LIVE_PMC.get(config)
    if live is not None:
        return live, "rocprofv3 --pmc child runs of this command, in this run"
    try:
        pmc = json.load(open(os.path.join(ROOT, rel)))
    except Exception:
        return None, "no PMC summary for this command"
    sha = kernel_source_sha16()
    if pmc.get("source_sha16") != sha:
        return None, f"{rel} is stale (kernel sources {pmc.get('source_sha16')} != {sha}): not quoted"
    return pmc, rel


def roofline_record(config, brute, kernel_name, phase_ms, launch_ms, n_launches, device_ms, executed_cells, useful_laneops,
                    useful_cells, algorithmic_cells, algorithmic_bytes, extra=None, share=1.0):
    """One flat record (the driver keeps the scalars of `roofline`): the scoring kernels against the integer-VALU roof.
    share < 1 (N > 1, a sharded workload): this rank's part of the job's executed cells -- the counters were counted on
    the whole workload on one GPU and are scaled to what this rank's GPU ran (cells, times and launches are the rank's own)."""
    kernel_s = phase_ms / 1e3
    pmc, source = pmc_counters(config, brute)
    priced = executed_cells * LANEOPS_PER_CELL / kernel_s / 1e12
    rec = {"bound": "valu", "achieved": None, "peak": VALU_PEAK_TLANEOPS, "unit": "Tlane-op/s", "frac": None,
           "frac_priced": priced / VALU_PEAK_TLANEOPS,
           "frac_useful": useful_laneops / kernel_s / 1e12 / VALU_PEAK_TLANEOPS,
           "traffic": None, "kernel": kernel_name, "kernel_ms_per_step": phase_ms,
           "sum_of_launch_durations_ms": launch_ms, "n_launches_per_step": n_launches, "device_ms_per_step": device_ms,
           "valu_wave_instructions_per_step": None, "executed_cells_per_step": executed_cells,
           "useful_cells_per_step": useful_cells, "algorithmic_cells_per_step": algorithmic_cells,
           "executed_Tcell_per_s": executed_cells / kernel_s / 1e12,
           "algorithmic_bytes_per_step": algorithmic_bytes, "algorithmic_GBps": algorithmic_bytes / kernel_s / 1e9,
           "hbm_peak_GBps": HBM_PEAK_GBPS, "counters": source}
    if pmc is not None:
        scaled = lambda v: None if v is None else v * share
        valu = pmc["sweep_kernels"]["valu_wave_instructions_per_step"] * share
        issued = valu * 64.0 / kernel_s / 1e12              # lane-op slots the issued VALU instructions fill per second
        rec.update(achieved=issued, frac=issued / VALU_PEAK_TLANEOPS, valu_wave_instructions_per_step=valu,
                   traffic=scaled(pmc.get("hbm_bytes_per_step_sweep_kernels")),
                   fetch_bytes_per_step=scaled(pmc["sweep_kernels"].get("fetch_bytes")),
                   write_bytes_per_step=scaled(pmc["sweep_kernels"].get("write_bytes")))
        if share != 1.0:
            rec["counters"] = f"{source}; scaled to this rank's {share:.4f} of the job's executed cells" 
        if rec["traffic"]:
            rec["traffic_GBps"] = rec["traffic"] / kernel_s / 1e9
        cyc, clock = pmc.get("simd_cycles_per_valu_instruction_active"), pmc.get("clock_GHz")
        if cyc and clock:
            # a SIMD that issues one wave instruction (64 lanes) every `cyc` cycles at `clock`, against 32 lanes/clk at 2.4 GHz
            ceiling = (64.0 / cyc) / 32.0 * clock / 2.4
            rec.update(cycles_per_valu_instruction=cyc, clock_GHz=clock, issue_ceiling_frac=ceiling,
                       frac_over_issue_ceiling=rec["frac"] / ceiling)
    if extra:
        rec.update(extra)
    return rec


def stats_delta(st0, st1):
    """Per-run averages of the batch statistics between two snapshots (HIP events on the streams the kernels run on)."""
    runs = max(st1["n_runs"] - st0.get("n_runs", 0), 1)
    d = lambda k: (st1[k] - st0.get(k, 0.0)) / runs
    return dict(runs=runs, phase_ms=d("sum_score_phase_ms"), launch_ms=d("sum_score_kernel_ms"), total_ms=d("sum_total_ms"),
                ext_ms=d("sum_extent_kernel_ms"))


def init_dist(args):
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU path)")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    # NRA_BENCH_FORCE_DIST: rehearse the process-group path (RCCL init, all_reduce, all_gather, barrier) with
    # the one rank a one-GPU box allows; needs the launcher's MASTER_ADDR / MASTER_PORT / RANK / WORLD_SIZE
    if world > 1 or os.environ.get("NRA_BENCH_FORCE_DIST"):
        import torch.distributed as dist
        if not dist.is_initialized():
            if args.backend == "nccl":
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(backend=args.backend)
    return rank, local_rank, world, dist


def timed_steps(args, dist, step, on_warm=None, on_done=None):
    """W warm-up steps, then exactly K steps (+ on_done, still inside) between barrier + synchronize; max over ranks."""
    import torch

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    if on_warm is not None:
        on_warm()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if on_done is not None:
        on_done()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        dev = torch.device("cuda", torch.cuda.current_device()) if args.backend == "nccl" else torch.device("cpu")
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


VALUE_DEFINITION = ("value: inputs resident in HBM (the contract); ms_per_scorer_call / value_scorer_call: one call from host "
                    "buffers to host results (SURVEY 8d), median; cpu ratio uses it")


def bench_1d(args):
    """Configs 2, 4, 5.  Returns the record (rank 0) or None."""
    import numpy as np
    from nanorepeat_amd import _capi as A, dist as D, synth
    rank, local_rank, world, dist = init_dist(args)

    if args.config in (2, 5):
        # every rank owns one region of `reads` reads (weak scaling); rank 0's is the BASELINE config exactly
        make = synth.config2 if args.config == 2 else synth.config5
        data = make(n_reads=args.reads, seed=synth.SEED + rank)
        index = rank * args.reads + np.arange(args.reads, dtype=np.int64)
        n_total = world * args.reads
        workload = ("config2: 10k synthetic ONT-error core reads (q~400/950), motif TATTG, k in [5,200] "
                    "(196 candidates/read), 1000 bp flanks" if args.config == 2 else
                    f"config5: {args.reads} HiFi-error core reads (q~500/2300), TATTG, k in [5,500] (496 candidates/read)")
        workload += "" if world == 1 else f"; one such region per GPU x {world}"
        scaling = "weak"
    else:
        # the whole job is fixed; regions go to ranks by the cells the kernels will execute for them,
        # computed from the region descriptors alone, and every rank materialises only its own reads
        cost = np.array([synth.config4_region_cost(synth.config4_region(g), args.reads_per_region)
                         for g in range(args.regions)], np.int64)
        owner = D.lpt_assign(cost, world)
        data = synth.config4(args.regions, args.reads_per_region, only=np.nonzero(owner == rank)[0])
        index = data["read_id"]
        n_total = args.regions * args.reads_per_region
        workload = (f"config4: {args.regions} regions x {args.reads_per_region} reads, 3-6 bp motifs, ont_q20 errors, reference "
                    f"window rule (K ~ 31), regions dealt over {world} GPU(s)")
        scaling = "strong"

    n_align_local = int(np.maximum(data["kmax"].astype(np.int64) - data["kmin"] + 1, 0).sum())
    sb = D.ShardedBatch1D(data["regions"], data["reads"], data["kmin"], data["kmax"], data.get("read_region"),
                          index, n_total, flags=(A.F_BRUTE_FORCE if args.brute else 0) | (A.F_NO_QUANTA if args.no_quanta else 0),
                          device=local_rank)
    last = {}

    # One step = one pass over the rank's shard: kernels -> on-device selection -> D2H of the per-read results
    # -> (N > 1) the all_gather.  The exchange of pass i runs on the host / RCCL while the kernels of pass
    # i + 1 execute; the last one is exposed before the closing barrier.  N = 1: nothing to overlap.
    def step():
        sb.run()
        if "pending" in last:
            last["out"] = sb.exchange(last.pop("pending"))
        last["pending"] = sb.fetch_local()

    def drain():
        if "pending" in last:
            last["out"] = sb.exchange(last.pop("pending"))

    warm = {}
    dt = timed_steps(args, dist, step, on_warm=lambda: (drain(), warm.update(sb.stats())), on_done=drain)
    st = sb.stats()
    out = last["out"]
    n_align = n_align_local
    if dist is not None:
        import torch
        dev = torch.device("cuda", torch.cuda.current_device()) if args.backend == "nccl" else torch.device("cpu")
        t = torch.tensor([n_align_local], dtype=torch.int64, device=dev)
        dist.all_reduce(t)
        n_align = int(t.item())

    line = None
    dl = stats_delta(warm, st)
    multi = None
    job_share = 1.0
    if dist is not None:
        # one more pass with NOTHING behind it: the exchange (one all_gather of 32 B per read, padded to the largest shard)
        # exposed, timed between barriers; then every rank's kernel time and executed cells
        import torch
        dev = torch.device("cuda", torch.cuda.current_device()) if args.backend == "nccl" else torch.device("cpu")
        sb.run()
        local = sb.fetch_local()
        dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        sb.exchange(local)
        dist.barrier()
        t_gather = time.perf_counter() - t0
        mine_t = torch.tensor([dl["phase_ms"], float(st["executed_cells"]), t_gather * 1e3, float(n_align_local)], dtype=torch.float64, device=dev)
        allr = [torch.empty_like(mine_t) for _ in range(world)]
        dist.all_gather(allr, mine_t)
        allr = np.array([t.cpu().numpy() for t in allr])
        if scaling == "strong":        # the counters were counted on the whole job on one GPU: this rank ran its share of it
            job_share = float(st["executed_cells"]) / max(float(allr[:, 1].sum()), 1.0)
        multi = {"per_rank_kernel_ms": [round(float(x), 3) for x in allr[:, 0]],
                 "per_rank_alignments": [int(x) for x in allr[:, 3]],
                 "executed_cells_max_over_mean": float(allr[:, 1].max() / max(allr[:, 1].mean(), 1.0)),
                 "exposed_all_gather_ms": float(allr[:, 2].max()),
                 "note": "exposed_all_gather_ms: the exchange of ONE pass with no next pass behind it (in the timed steps pass i's "
                         "exchange runs beside pass i+1's kernels)"}
    if rank == 0:
        mine = out["status"][index] == 0
        est = out["sum_k"][index][mine] / np.maximum(out["n_ties"][index][mine], 1)
        exact = float(np.mean(est == data["k_true"][mine])) if mine.any() else 0.0
        kern = ("k_score_pk16<R>" if args.brute else
                "k_sweep_ring<R,dir> / k_sweep_ring32<R,dir>" +
                (" + k_sweep_ringmt<R,dir>" if max(len(r) for r in data["reads"]) > 1536 else ""))
        # the decomposition's cells without row padding and pipeline fill: per read q x (|L| + m kmax + |R|), packed int16
        qlen = np.fromiter((len(r) for r in data["reads"]), np.int64, len(data["reads"]))
        rr = data.get("read_region")
        reg_of = np.zeros(len(qlen), np.int64) if rr is None else np.asarray(rr, np.int64)
        flank = np.array([len(g[0]) + len(g[2]) for g in data["regions"]], np.int64)[reg_of]
        unit = np.array([len(g[1]) for g in data["regions"]], np.int64)[reg_of]
        useful = int((qlen * (flank + unit * np.maximum(data["kmax"].astype(np.int64), 0)))[data["kmax"] >= data["kmin"]].sum())
        if args.brute:
            useful = int(st["algorithmic_cells"])
        line = {
            "metric": "read-alignments/sec (reads x candidate-k)",
            "value": n_align * args.steps / dt, "unit": "read-alignments/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "value_scorer_call": None, "ms_per_scorer_call": None,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "int16", "data": "synthetic",
            "config": {"workload": workload, "reads": n_total, "alignments": n_align,
                       "mode": "brute force (K independent alignments)" if args.brute else "junction decomposition (exact)",
                       "timed_region": "inputs resident in HBM; kernels + on-device selection + D2H of per-read results" +
                                       (" + all_gather" if world > 1 else ""),
                       "parallelism": f"region blocks over {world} GPU(s), no data-path collective, one all_gather of 32 B/read"},
            "roofline": roofline_record(args.config, args.brute, kern, dl["phase_ms"], dl["launch_ms"], st["n_score_launches"],
                                        dl["total_ms"], st["executed_cells"], useful * LANEOPS_PACKED_CELL, useful,
                                        st["algorithmic_cells"], st["algorithmic_bytes"],
                                        {"extent_kernel_ms_per_step": dl["ext_ms"], "junction_snapshot_bytes_per_step": st["intermediate_bytes"]},
                                        share=job_share),
            "extent_tasks_per_step": st["n_extent_tasks"],
            "exact_k_fraction": exact,
        }
        if multi is not None:
            line["multi_gpu"] = multi
    sb.close()
    if rank == 0 and world == 1 and args.one_shot_calls > 0 and not args.brute:
        # ONE nra_round3_1d call, host buffers (ASCII reads) in, host results out
        call, res = A.prepared_round3_1d(data["regions"], data["reads"], data["kmin"], data["kmax"],
                                         read_region=data.get("read_region"), device=local_rank)
        call()
        ts = []
        for _ in range(args.one_shot_calls):
            t0 = time.perf_counter(); call(); ts.append(time.perf_counter() - t0)
        med = float(np.median(ts))
        same = all(np.array_equal(res[k], out[k][index]) for k in ("best_score", "sum_k", "n_ties", "status"))
        line["value_scorer_call"] = n_align / med
        line["ms_per_scorer_call"] = med * 1e3
        line["scorer_call"] = {"calls": args.one_shot_calls, "equals_resident_results": bool(same),
                               "what": "one nra_round3_1d call, host buffers (ASCII reads) -> host results; median"}
    if rank == 0 and world == 1 and args.cpu_sample != 0:
        cb = cpu_baseline_1d(data, args.cpu_sample, getattr(args, "cpu_seconds", 15.0))
        if cb is not None:
            base, ref = cb
            n = len(ref["sum_k"])
            same = all(np.array_equal(out[k][index][:n], ref[k]) for k in ("sum_k", "n_ties", "status", "best_score"))
            base["gpu_matches_sample"] = bool(same)
            if "_decomposed" in base:
                m, dec = base.pop("_decomposed")
                base["gpu_matches_same_algorithm_sample"] = bool(all(np.array_equal(out[k][index][:m], dec[k]) for k in ("sum_k", "n_ties", "status", "best_score")))
                if line["value_scorer_call"]:
                    base["gpu_over_cpu_same_algorithm"] = line["value_scorer_call"] / base["same_algorithm_value"]
            if line["value_scorer_call"]:
                base["gpu_over_cpu"] = line["value_scorer_call"] / base["value"]
            line["cpu_baseline"] = base
    if dist is not None:
        dist.barrier()
    return line


def bench_joint(args):
    """BASELINE config 3: the two grid rounds of the joint mode (nanoRepeat_joint.py:266-269) on 5000
    HTT-like amplicon reads, through the product host path (joint.fine_tune_read_count) on a resident
    GridSession.  One step = round 2 + round 3: per round the grid routing of the reference, the kernels and the
    fetch of the per-read results.  The scorer call of SURVEY 8(d) = the same from the reads as host strings
    (packing + H2D of the reads + both rounds)."""
    import numpy as np
    from nanorepeat_amd import joint as J, synth
    rank, local_rank, world, dist = init_dist(args)
    if world != 1:
        raise SystemExit("bench.py --config 3 is a one-GPU workload (5000 reads); see dist.joint_2d_sharded for N > 1")
    n = args.reads if args.reads != 10000 else 5000
    j = synth.config3(n)
    init = J.Round1Estimation()
    fq = {}
    for i, s in enumerate(j["reads"]):
        init.repeat1_count_range_dict[f"r{i}"] = tuple(int(x) for x in j["range1"][i])
        init.repeat2_count_range_dict[f"r{i}"] = tuple(int(x) for x in j["range2"][i])
        init.read_strand_dict[f"r{i}"] = int(j["strand"][i])        # round 1 knows the strand: no probe in the grid rounds
        fq[f"r{i}"] = f"@r{i}\n{s}\n+\n{'!' * len(s)}\n"
    left, u1, mid, u2, right = j["region"]
    chrom = left + u1 * 19 + mid + u2 * 7 + right
    a = J.Repeat.parse(f"chr4:{len(left)}:{len(left) + 57}:{u1}:200")
    b = J.Repeat.parse(f"chr4:{len(left) + 57 + len(mid)}:{len(left) + 57 + len(mid) + 21}:{u2}:20")
    a.max_size += 10; b.max_size += 10                               # nanoRepeat_joint.py:202-203
    from nanorepeat_amd import _capi as A
    session = J.GridSession(J._joint_region(chrom, a, b), fq, device=local_rank, flags=(A.F_JOINT_TAILS if args.joint_tails else 0) | (A.F_JOINT_NO_CHAIN if args.joint_no_chain else 0) |
                                  (A.F_JOINT_NO_KEEP if args.joint_no_keep else 0))
    last = {}

    def step(sess=session):
        sess.new_run()          # every step is a whole run: round 2 makes the reverse sweeps, round 3 reuses them
        last["est"] = J.fine_tune_read_count(init, fq, chrom, copy.copy(a), copy.copy(b), device=local_rank, session=sess)

    for _ in range(args.warmup):
        step()
    session.rounds = []
    dt = timed_steps(argparse.Namespace(**dict(vars(args), warmup=0)), dist, step)
    rounds = session.rounds
    session.rounds = None
    # statistics: every resident batch (one per read group) accumulates its HIP-event times over its runs
    n_cells = sum(c for c, _, _ in rounds) // args.steps
    exe = sum(st["executed_cells"] for _, st, _ in rounds) // args.steps
    alg = sum(st["algorithmic_cells"] for _, st, _ in rounds) // args.steps
    cells_per_round = {}
    phase_ms = total_ms = 0.0
    for key in sorted({k for _, _, k in rounds}):
        mine = [(c, st) for c, st, k in rounds if k == key]
        first, final = mine[0][1], mine[-1][1]
        phase_ms += (final["sum_score_phase_ms"] - first["sum_score_phase_ms"] + first["score_phase_ms"]) / args.steps
        total_ms += (final["sum_total_ms"] - first["sum_total_ms"] + first["total_ms"]) / args.steps
        per_step = len(mine) // args.steps
        for i, (c, _) in enumerate(mine[:per_step]):
            cells_per_round[i] = cells_per_round.get(i, 0) + c
    n_groups = len({k for _, _, k in rounds})
    est = last["est"]
    k1 = np.array([est.repeat1_count_dict.get(f"r{i}", -1) for i in range(n)])
    k2 = np.array([est.repeat2_count_dict.get(f"r{i}", -1) for i in range(n)])
    ms_step = dt / args.steps * 1e3
    n_launches = sum(st["n_score_launches"] for _, st, _ in rounds) // args.steps
    launch_ms = sum(st["score_kernel_ms"] for _, st, _ in rounds) / args.steps
    # The decomposition's cells without row padding and pipeline drain (an estimate from the reads and their ranges):
    # packed int16 flank columns; int32 window columns of the prefix sweep (to the last kept k1) and of the extended reverse
    # sweep (to the last kept k2); MID scans of 1 + |mid| columns per scored (read, k1) -- round 2's grid values inside the
    # read's range, round 3's 2 x step counts
    q = np.fromiter((len(r) for r in j["reads"]), np.int64, n)
    l1, m1, l2, m2, l3 = len(left), len(u1), len(mid), len(u2), len(right)
    cl, cr_ = max(l1 - 10, 0), max(l3 - 10, 0)
    r1, r2 = np.asarray(j["range1"], np.int64), np.asarray(j["range2"], np.int64)
    s1, s2 = max(est.step_size1, 1), max(est.step_size2, 1)
    steps2 = [J.choose_best_step_size(rep, d) for rep, d in ((a, init.repeat1_count_range_dict), (b, init.repeat2_count_range_dict))]
    n1_r2 = np.maximum((r1[:, 1] - r1[:, 0] + steps2[0] - 1) // steps2[0], 1)
    n1_r3 = np.minimum(2 * steps2[0], r1[:, 1] - r1[:, 0]) if min(steps2) > 1 else 0
    useful_packed = int((q * (cl + cr_)).sum())
    useful_int32 = int((q * ((l1 - cl) + m1 * (r1[:, 1] - 1) + (l3 - cr_) + m2 * (r2[:, 1] - 1) + (1 + l2) * (n1_r2 + n1_r3))).sum())
    line = {
        "metric": "read-alignments/sec (reads x candidate cells)",
        "value": n_cells * args.steps / dt, "unit": "read-alignments/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
        "value_scorer_call": None, "ms_per_scorer_call": None,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
        "config": {"workload": f"config3: HTT-like joint CAG+CCG grid rounds 2+3, {n} amplicon reads (1.2 kb, either strand), "
                               "ranges [k-20, k+5)", "reads": n, "alignments": n_cells, "read_groups": n_groups,
                   "round3_on_device": bool(getattr(est, "refined", False)),
                   "timed_region": "reads resident in HBM; flank sweeps ahead of the grids, round 2, round 3 routed on the "
                                   "device, one D2H of per-read results, the reference's result dicts"},
        "roofline": roofline_record(3, False, "k_joint_pk16<R> + k_joint_sweep<R,dir> + k_joint_midscan<R> + k_joint_combine",
                                    phase_ms, launch_ms, n_launches, total_ms, exe,
                                    useful_packed * LANEOPS_PACKED_CELL + useful_int32 * LANEOPS_INT32_CELL,
                                    useful_packed + useful_int32, alg,
                                    int(sum(st["algorithmic_bytes"] for _, st, _ in rounds) // args.steps),
                                    {"host_ms_per_step": ms_step - total_ms, "useful_cells_packed": useful_packed,
                                     "useful_cells_int32": useful_int32}),
        "k1_within1": float(np.mean(np.abs(k1 - j["truth"][:, 0]) <= 1)),
        "k2_within1": float(np.mean(np.abs(k2 - j["truth"][:, 1]) <= 1)),
    }
    session.close()
    if args.one_shot_calls > 0:
        # the whole scorer call from the reads as host strings: packing + H2D + both grid rounds
        def call():
            t0 = time.perf_counter()
            got = J.fine_tune_read_count(init, fq, chrom, copy.copy(a), copy.copy(b), device=local_rank)
            return time.perf_counter() - t0, got
        call()
        ts, got = [], None
        for _ in range(args.one_shot_calls):
            t, got = call()
            ts.append(t)
        med = float(np.median(ts))
        line["value_scorer_call"] = n_cells / med
        line["ms_per_scorer_call"] = med * 1e3
        line["scorer_call"] = {"calls": args.one_shot_calls,
                               "equals_resident_results": bool(got.repeat1_count_dict == est.repeat1_count_dict and
                                                               got.repeat2_count_dict == est.repeat2_count_dict),
                               "what": "joint.fine_tune_read_count from the FASTQ dict (packing + H2D + both rounds); median"}
    if args.cpu_sample != 0:
        from oracle import oracle as O
        cores = host_cores()
        secs = getattr(args, "cpu_seconds", 10.0)
        m = min(n, int(6.4 * secs * cores)) if args.cpu_sample < 0 else min(args.cpu_sample, n)
        # a coarse routed grid (step 7 on both axes) over the first m reads' round-1 ranges: the oracle cell by cell, and the
        # product path on the same grid (routing in the library, junction at the end of mid) for the comparison
        r1, r2 = np.asarray(j["range1"][:m], np.float64), np.asarray(j["range2"][:m], np.float64)
        grid = A.Grid((0, 7, int(r1[:, 1].max()) // 7 + 1), r1[:, 0], r1[:, 1], (0, 7, int(r2[:, 1].max()) // 7 + 1), r2[:, 0], r2[:, 1])
        cr, c1, c2 = A.joint_grid_cells(grid)
        strands_m = np.asarray(j["strand"][:m], np.int8)
        t0 = time.perf_counter()
        want = O.joint_2d(j["region"], j["reads"][:m], cr, c1, c2, read_strand=strands_m, threads=cores)
        dtc = time.perf_counter() - t0
        with A.Batch.create_2d_reads(j["region"], j["reads"][:m], device=local_rank) as sample_batch:
            sample_batch.set_grid(grid, strands_m)
            sample_batch.run(); sample_batch.sync()
            have = sample_batch.fetch()
        listed = np.zeros(m, bool); listed[cr] = True
        same = all(np.array_equal(np.asarray(have[k])[listed if len(want[k]) == m else slice(None)],
                                  np.asarray(want[k])[listed if len(want[k]) == m else slice(None)]) for k in want)
        line["cpu_baseline"] = {"value": len(cr) / dtc, "unit": "read-alignments/s", "cores": cores, "kind": "port",
                                "sample": f"first {m} reads x a step-7 grid over their ranges = {len(cr)} cells in {dtc:.1f} s; "
                                          "oracle/nr_oracle.c (one optimal DP with window payload per cell), not minimap2",
                                "gpu_matches_sample": bool(same)}
        if line["value_scorer_call"]:
            line["cpu_baseline"]["gpu_over_cpu"] = line["value_scorer_call"] / line["cpu_baseline"]["value"]
    return line


def sub_record(args, config):
    """One of the other BASELINE workloads for the default run's `configs`: a few steps, a shorter CPU sample.  N > 1: config 4
    only, its regions dealt over all the ranks (strong scaling; every rank calls this)."""
    sub = argparse.Namespace(**vars(args))
    sub.config = config
    sub.steps, sub.warmup = {3: (12, 2), 4: (2, 1), 5: (5, 1)}[config]      # (config 3's 6 ms steps are a tenth host: three of them are noise)
    sub.reads = 10000
    sub.one_shot_calls = min(args.one_shot_calls, 2 if config == 4 else 3)
    sub.cpu_seconds = 6.0
    t0 = time.perf_counter()
    try:
        rec = bench_joint(sub) if config == 3 else bench_1d(sub)
    except Exception as e:          # a sub-record must not take the headline down with it
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            raise                   # (N > 1: a rank that went on alone would hang the others in their next collective)
        return {"error": f"{type(e).__name__}: {e}"}
    if rec is not None:
        rec["wall_s_of_this_record"] = round(time.perf_counter() - t0, 1)
        # what a sub-record shares with the headline is not repeated: the whole line has to fit the 8 KB the driver keeps
        for k in ("metric", "unit", "higher_is_better", "vs_baseline", "data", "value_scorer_call"):
            rec.pop(k, None)
        for k in ("mode", "timed_region", "parallelism"):
            rec["config"].pop(k, None)
        for k in ("bound", "peak", "unit", "hbm_peak_GBps", "algorithmic_GBps", "executed_Tcell_per_s", "extent_kernel_ms_per_step",
                  "junction_snapshot_bytes_per_step", "traffic_GBps", "issue_ceiling_frac", "sum_of_launch_durations_ms"):
            rec["roofline"].pop(k, None)
        if "cpu_baseline" in rec:
            rec["cpu_baseline"]["sample"] = rec["cpu_baseline"]["sample"].split(";")[0]
            rec["cpu_baseline"].pop("same_algorithm_what", None)
        if "scorer_call" in rec:
            rec["scorer_call"].pop("what", None)
    return rec


def compact(x):
    """Numbers at 5 significant digits: the line has to fit the 8 KB the driver keeps."""
    if isinstance(x, float):
        return float(f"{x:.5g}")
    if isinstance(x, dict):
        return {k: compact(v) for k, v in x.items()}
    if isinstance(x, list):
        return [compact(v) for v in x]
    return x


def main():
    args = parse()
    spawn_ranks_if_needed(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    subs = [] if args.sub_configs in ("none", "") else [int(x) for x in args.sub_configs.split(",") if int(x) in (3, 4, 5)]
    default_run = world == 1 and args.config == 2 and not args.brute
    if not default_run:
        subs = [4] if (world > 1 and args.config == 2 and not args.brute and 4 in subs) else []
    if (args.live_pmc == "on" or (args.live_pmc == "auto" and default_run and subs)) and world == 1:
        # before this process touches the GPU: the counters of this very build, one config after the other
        for c in [args.config] + subs:
            extra = ["--reads", str(args.reads)] if c == args.config and c != 4 else []
            got = live_pmc(c, PMC_KERNELS[c], steps=PMC_STEPS[c], extra=extra)
            if got is not None:
                LIVE_PMC[c] = got
    line = bench_joint(args) if args.config == 3 else bench_1d(args)
    configs = {}
    for c in subs:
        configs[f"config{c}"] = sub_record(args, c)
    if line is not None:
        line["value_definition"] = VALUE_DEFINITION
        line["notes"] = "DESIGN.md 5 explains every field; profiles/r04_* hold the rocprofv3 summaries"
        if configs:
            line["configs"] = configs
        print(json.dumps(compact(line), separators=(",", ":")), flush=True)
    if world > 1:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()


if __name__ == "__main__":
    main()

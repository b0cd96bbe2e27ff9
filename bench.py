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

`value` is that resident-input rate (the contract of this benchmark: inputs in HBM when the clock
starts).  SURVEY.md 8(d) defines the metric on the wall time of the scorer call from host buffers
to host results; that figure is measured in the same run and reported beside it as `one_shot`
(N = 1, config 2: one nra_round3_1d call = 2-bit packing + device buffers + H2D + kernels +
selection + D2H, median of --one-shot-calls calls).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|3|4]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import subprocess
import sys
import time

# before torch starts the HIP runtime: nanorepeat_amd.RECOMMENDED_ENV (an entry point's choice, not the library's)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Integer-VALU roofline of the dominant kernel (DESIGN.md "Roofline"):
#   peak lane-ops/s = 256 CU x 4 SIMD x 32 lanes/clk x 2.4 GHz
#   a two-piece-affine local-alignment cell is priced at 20 packed-int16 VALU instructions per
#   2 cells = 10 lane-ops per cell (SURVEY.md 8d's 8-13); `achieved` counts the cells the kernels
#   EXECUTE (padding and pipeline fill included) at that price.
VALU_PEAK_TLANEOPS = 256 * 4 * 32 * 2.4e9 / 1e12      # 78.6
LANEOPS_PER_CELL = 10.0
HBM_PEAK_GBPS = 8000.0
PMC_PROFILE = os.path.join("profiles", "r02_pmc_traffic.json")


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
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL)")
    ap.add_argument("--same-device", action="store_true",
                    help="rehearsal only: every rank uses GPU 0 (with --backend gloo on a one-GPU box)")
    ap.add_argument("--cpu-sample", type=int, default=-1,
                    help="reads in the CPU-baseline sample (-1: sized for ~15 s, 0: skip)")
    ap.add_argument("--one-shot-calls", type=int, default=5, help="host-buffers-in/out calls timed at N = 1 (0: skip)")
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


def cpu_baseline_1d(data, n_sample):
    """Times the CPU oracle (oracle/, the restatement of the reference algorithm: K independent
    optimal alignments per read) on a bounded sample of the same workload, all host cores."""
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
        n_sample = int(n_cal * 15.0 / t_cal)
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
    return {"value": n_align / dt, "unit": "read-alignments/s", "cores": cores, "kind": "port",
            "sample": f"first {n_sample} of the workload's reads = {n_align} alignments in {dt:.1f} s; CPU "
                      f"restatement (optimal DP, K independent alignments per read, oracle/nr_oracle.c, "
                      f"OpenMP over reads), not minimap2"}, out


def roofline(st0, st1, n_steps, brute, kernel_name):
    """The dominant kernels against the integer-VALU roof, from the HIP events of the timed steps
    (st0/st1: batch statistics before/after them; the events sit on the streams the kernels run on)."""
    runs = st1["n_runs"] - st0["n_runs"]
    phase_ms = (st1["sum_score_phase_ms"] - st0["sum_score_phase_ms"]) / max(runs, 1)
    launch_ms = (st1["sum_score_kernel_ms"] - st0["sum_score_kernel_ms"]) / max(runs, 1)
    total_ms = (st1["sum_total_ms"] - st0["sum_total_ms"]) / max(runs, 1)
    ext_ms = (st1["sum_extent_kernel_ms"] - st0["sum_extent_kernel_ms"]) / max(runs, 1)
    kernel_s = phase_ms / 1e3
    exe_cells_per_s = st1["executed_cells"] / kernel_s
    alg_cells_per_s = st1["algorithmic_cells"] / kernel_s
    achieved = exe_cells_per_s * LANEOPS_PER_CELL / 1e12
    traffic, counters = None, None
    prof = os.path.join(ROOT, PMC_PROFILE)
    if os.path.exists(prof) and not brute:
        try:
            pmc = json.load(open(prof))
            traffic = pmc.get("hbm_bytes_per_step_sweep_kernels")
            counters = {"source": PMC_PROFILE + " (rocprofv3 --pmc passes of this command)",
                        "valu_wave_instructions_per_step": pmc["sweep_kernels"]["valu_wave_instructions_per_step"],
                        "simd_cycles_per_valu_instruction": pmc.get("simd_cycles_per_valu_instruction")}
        except Exception:
            traffic, counters = None, None
    hbm_gbps = st1["algorithmic_bytes"] / kernel_s / 1e9
    return {"bound": "valu", "achieved": achieved, "peak": VALU_PEAK_TLANEOPS, "unit": "Tlane-op/s",
            "frac": achieved / VALU_PEAK_TLANEOPS, "traffic": traffic, "counters": counters,
            "kernel": kernel_name,
            "kernel_ms_per_step": phase_ms, "steps_averaged": runs,
            "sum_of_launch_durations_ms": launch_ms, "n_launches_per_step": st1["n_score_launches"],
            "executed_cells_per_step": st1["executed_cells"], "executed_Tcell_per_s": exe_cells_per_s / 1e12,
            "laneops_per_cell": LANEOPS_PER_CELL,
            "note": "achieved = cells the kernels execute (row padding to 64*R and pipeline fill/drain columns "
                    "included) x 10 lane-ops / the scoring phase's HIP-event time averaged over the timed steps "
                    "(launches of different read-length buckets overlap on their own streams). The K-fold "
                    "algorithmic cell count of SURVEY 8d is in 'algorithmic': the junction decomposition "
                    "shares L+unit^k and R across the K candidates and never executes those cells",
            "algorithmic": {"cells_per_step": st1["algorithmic_cells"], "Tcell_per_s": alg_cells_per_s / 1e12,
                            "over_executed": st1["algorithmic_cells"] / max(st1["executed_cells"], 1)},
            "hbm": {"algorithmic_bytes_per_step": st1["algorithmic_bytes"], "achieved_GBps": hbm_gbps,
                    "peak_GBps": HBM_PEAK_GBPS, "frac": hbm_gbps / HBM_PEAK_GBPS,
                    "junction_snapshot_bytes_per_step": st1["intermediate_bytes"],
                    "note": "compute-bound by design: ~5 B per read-alignment of inputs and results; 'traffic' (PMC) "
                            "also holds the decomposition's own hand-off between the reverse and the forward sweep "
                            "(the R side of the junction, 12 B per row pair, written once and read once)"},
            "extent_kernel_ms_per_step": ext_ms, "device_ms_per_step": total_ms}


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


def bench_1d(args):
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
                    f"config5: {args.reads} synthetic HiFi-error core reads (q~500/2300), motif TATTG, k in [5,500] "
                    "(496 candidates/read, wide sweep), 1000 bp flanks")
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
        workload = (f"config4: {args.regions} regions x {args.reads_per_region} reads, mixed 3-6 bp motifs, ont_q20 "
                    f"errors, reference window rule (K = 31 typical), sharded by region over {world} GPU(s)")
        scaling = "strong"

    n_align_local = int(np.maximum(data["kmax"].astype(np.int64) - data["kmin"] + 1, 0).sum())
    sb = D.ShardedBatch1D(data["regions"], data["reads"], data["kmin"], data["kmax"], data.get("read_region"),
                          index, n_total, flags=A.F_BRUTE_FORCE if args.brute else 0, device=local_rank)
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

    if rank == 0:
        mine = out["status"][index] == 0
        est = out["sum_k"][index][mine] / np.maximum(out["n_ties"][index][mine], 1)
        exact = float(np.mean(est == data["k_true"][mine])) if mine.any() else 0.0
        kern = "k_score_pk16<R>" if args.brute else "k_sweep_ring<R,dir> (reverse + forward sweeps of all read-length buckets)"
        line = {
            "metric": "read-alignments/sec (reads x candidate-k)",
            "value": n_align * args.steps / dt, "unit": "read-alignments/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "int16", "data": "synthetic",
            "config": {"workload": workload, "reads": n_total, "alignments": n_align,
                       "mode": "brute force (K independent alignments)" if args.brute else
                               "junction decomposition (exact; shares L+unit^k and R across k)",
                       "timed_region": "inputs resident in HBM; kernels + on-device selection + D2H of per-read "
                                       "results" + (" + all_gather (pass i's exchange overlaps pass i+1's kernels; the last "
                                                    "one is exposed)" if world > 1 else ""),
                       "parallelism": f"region blocks sharded over {world} GPU(s), no data-path collective, one all_gather of 32 B/read"},
            "roofline": roofline(warm, st, args.steps, args.brute, kern),
            "extent_tasks_per_step": st["n_extent_tasks"],
            "exact_k_fraction": exact,
        }
        if world == 1 and args.config == 2 and args.one_shot_calls > 0 and not args.brute:
            call, res = A.prepared_round3_1d(data["regions"], data["reads"], data["kmin"], data["kmax"], device=local_rank)
            call()
            ts = []
            for _ in range(args.one_shot_calls):
                t0 = time.perf_counter(); call(); ts.append(time.perf_counter() - t0)
            med = float(np.median(ts))
            same = all(np.array_equal(res[k], out[k]) for k in ("best_score", "sum_k", "n_ties", "status"))
            line["one_shot"] = {"value": n_align / med, "unit": "read-alignments/s", "ms_per_call": med * 1e3,
                                "calls": args.one_shot_calls, "ms_all": [t * 1e3 for t in ts],
                                "equals_resident_results": bool(same),
                                "what": "SURVEY 8(d) wall time of the scorer call: one nra_round3_1d from host buffers "
                                        "(ASCII reads) to host results = 2-bit packing + device arena + H2D + kernels + "
                                        "selection + D2H; median"}
        if world == 1 and args.cpu_sample != 0:
            cb = cpu_baseline_1d(data, args.cpu_sample)
            if cb is not None:
                base, ref = cb
                n = len(ref["sum_k"])
                same = all(np.array_equal(out[k][index][:n], ref[k]) for k in ("sum_k", "n_ties", "status", "best_score"))
                base["gpu_matches_sample"] = bool(same)
                line["cpu_baseline"] = base
        print(json.dumps(line), flush=True)
    sb.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def bench_joint(args):
    """BASELINE config 3: the two grid rounds of the joint mode (nanoRepeat_joint.py:266-269) on 5000
    HTT-like amplicon reads, through the product host path (joint.fine_tune_read_count) on a resident
    GridSession.  One step = round 2 + round 3: per round the host builds the cell list from the previous
    round's estimates, sets it on the resident reads, runs the kernels and fetches the per-read results."""
    import copy
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
    # the product's session: from 2000 reads on two groups of reads in parallel host threads, so that one group's
    # host work overlaps the other's kernels (joint.GridSession)
    session = J.GridSession(J._joint_region(chrom, a, b), fq, device=local_rank)
    last = {}

    def step(sess=session):
        sess.new_run()          # every step is a whole run: round 2 makes the reverse sweeps, round 3 reuses them
        last["est"] = J.fine_tune_read_count(init, fq, chrom, copy.copy(a), copy.copy(b), device=local_rank, session=sess)

    dt = timed_steps(args, dist, step)
    est_split = last["est"]
    # kernel statistics from an un-split session of the same reads, where the batch's HIP-event times are those of
    # one stream (in the timed steps the two groups' kernels overlap)
    serial = J.GridSession(J._joint_region(chrom, a, b), fq, device=local_rank, parts=1)
    for _ in range(2):
        step(serial)
    serial.rounds = []
    n_serial = 4
    t_serial = time.perf_counter()
    for _ in range(n_serial):
        step(serial)
    t_serial = (time.perf_counter() - t_serial) / n_serial * 1e3
    rounds = serial.rounds
    per_step = len(rounds) // n_serial
    n_cells = sum(c for c, _ in rounds[:per_step])
    # statistics: the batch accumulates event times over its runs; cells are per round
    first, final = rounds[0][1], rounds[-1][1]
    runs = final["n_runs"] - first["n_runs"] + 1
    phase_ms = (final["sum_score_phase_ms"] - first["sum_score_phase_ms"] + first["score_phase_ms"]) / runs * per_step
    total_ms = (final["sum_total_ms"] - first["sum_total_ms"] + first["total_ms"]) / runs * per_step
    exe = sum(st["executed_cells"] for _, st in rounds[:per_step])
    alg = sum(st["algorithmic_cells"] for _, st in rounds[:per_step])
    est = last["est"]
    k1 = np.array([est.repeat1_count_dict.get(f"r{i}", -1) for i in range(n)])
    k2 = np.array([est.repeat2_count_dict.get(f"r{i}", -1) for i in range(n)])
    achieved = exe * LANEOPS_PER_CELL / (phase_ms / 1e3) / 1e12
    line = {
        "metric": "read-alignments/sec (reads x candidate cells)",
        "value": n_cells * args.steps / dt, "unit": "read-alignments/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
        "config": {"workload": f"config3: HTT-like joint CAG+CCG grid rounds 2+3, {n} amplicon reads (1.2 kb, either strand), "
                               "round-1 ranges [k-20, k+5)", "reads": n, "alignments": n_cells,
                   "cells_per_round": [c for c, _ in rounds[:per_step]],
                   "timed_region": "reads resident in HBM; per round: host cell list -> nra_batch2d_set_cells -> kernels -> "
                                   "on-device selection -> D2H of per-read results"},
        "roofline": {"bound": "valu", "achieved": achieved, "peak": VALU_PEAK_TLANEOPS, "unit": "Tlane-op/s",
                     "frac": achieved / VALU_PEAK_TLANEOPS, "traffic": None,
                     "kernel": "k_joint_sweep<R,dir> (reverse, prefix and tail sweeps)",
                     "kernel_ms_per_step": phase_ms, "device_ms_per_step": total_ms,
                     "host_ms_per_step": t_serial - total_ms, "unsplit_ms_per_step": t_serial,
                     "read_groups": max(1, len(session.subs)),
                     "executed_cells_per_step": exe, "executed_Tcell_per_s": exe / (phase_ms / 1e3) / 1e12,
                     "laneops_per_cell": LANEOPS_PER_CELL,
                     "note": "int32 cells = (score << 16 | window score): one cell per lane-op slot, priced like the 1D "
                             "cell (10 lane-ops) against the same 78.6 T lane-op/s; the columns outside the scoring window "
                             "run in packed int16 cells (k_joint_pk16).  kernel / device / host times are those of an "
                             "un-split session (unsplit_ms_per_step); ms_per_step is the product path, two read groups "
                             "in parallel host threads",
                     "algorithmic": {"cells_per_step": alg, "over_executed": alg / max(exe, 1)}},
        "read_groups_equal_unsplit": bool(est_split.repeat1_count_dict == est.repeat1_count_dict and
                                          est_split.repeat2_count_dict == est.repeat2_count_dict and
                                          list(est_split.repeat1_count_dict) == list(est.repeat1_count_dict)),
        "k1_within1": float(np.mean(np.abs(k1 - j["truth"][:, 0]) <= 1)),
        "k2_within1": float(np.mean(np.abs(k2 - j["truth"][:, 1]) <= 1)),
    }
    if args.cpu_sample != 0:
        from oracle import oracle as O
        cores = host_cores()
        m = min(n, 64 * cores) if args.cpu_sample < 0 else min(args.cpu_sample, n)      # ~10 s of CPU work
        cr, c1, c2 = [], [], []
        for r in range(m):
            for x in range(int(j["range1"][r][0]), int(j["range1"][r][1]), 7):
                for y in range(int(j["range2"][r][0]), int(j["range2"][r][1]), 7):
                    cr.append(r); c1.append(x); c2.append(y)
        t0 = time.perf_counter()
        O.joint_2d(j["region"], j["reads"][:m], cr, c1, c2, threads=cores)
        dtc = time.perf_counter() - t0
        line["cpu_baseline"] = {"value": len(cr) / dtc, "unit": "read-alignments/s", "cores": cores, "kind": "port",
                                "sample": f"first {m} reads x their step-7 grid = {len(cr)} cells in {dtc:.1f} s; CPU "
                                          "restatement (one optimal DP with window payload per cell), not minimap2"}
    print(json.dumps(line), flush=True)
    session.close()
    serial.close()


def main():
    args = parse()
    spawn_ranks_if_needed(args)
    if args.config == 3:
        return bench_joint(args)
    bench_1d(args)


if __name__ == "__main__":
    main()

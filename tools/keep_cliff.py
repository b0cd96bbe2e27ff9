"""Why do 32 GB of kept column states lose where 16 / 24 GB gain?  bench.py --config 3 --reads N with the
budget lifted (NRA_JOINT_KEEP_BUDGET_GB), N growing through the cliff:
  1. plain runs: step and device time, kept and not kept;
  2. rocprofv3 --kernel-trace (kernels concurrent, as in the product): time per kernel name;
  3. rocprofv3 --pmc passes (kernels serialised by the profiler): address-translation and write-request counters per kernel.
python3 tools/keep_cliff.py [out.json] [reads ...]          (on the GPU box; ~6 min)"""
import csv, glob, json, os, shutil, subprocess, sys, tempfile, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/keep_cliff.json"
sizes = [int(x) for x in sys.argv[2:]] or [20000, 30000, 40000]
BENCH = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "3", "--warmup", "1", "--cpu-sample", "0", "--one-shot-calls", "0",
         "--sub-configs", "none", "--live-pmc", "off"]
env = dict(os.environ, NRA_JOINT_KEEP_BUDGET_GB="64", TMPDIR="/tmp")
res = {"what": "config 3 with reads x 4 / x 6 / x 8: 16 / 24 / 32 GB of kept column states (budget lifted to 64 GB)", "runs": {}}


def short(name):
    return name.split("(")[0].replace("void ", "")


def plain(n, extra=()):
    r = subprocess.run(BENCH + ["--reads", str(n), "--steps", "4"] + list(extra), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=400)
    if r.returncode != 0:
        return {"error": r.stderr.decode()[-400:]}
    line = json.loads(r.stdout.decode().strip().splitlines()[-1])
    rf = line["roofline"]
    return {"ms_per_step": line["ms_per_step"], "device_ms_per_step": rf.get("device_ms_per_step"), "kernel_ms_per_step": rf.get("kernel_ms_per_step")}


def profiled(n, counters, steps=2, extra=()):
    out = tempfile.mkdtemp(prefix="nra_cliff_", dir="/tmp")
    cmd = ["rocprofv3"] + (["--pmc"] + counters if counters else []) + ["--kernel-trace", "--output-format", "csv", "-d", out, "--"] + \
          BENCH + ["--reads", str(n), "--steps", str(steps)] + list(extra)
    try:
        r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=500)
        if r.returncode != 0:
            return {"error": r.stderr.decode()[-400:]}
        per = {}
        if counters:
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                seen = set()
                for row in csv.DictReader(open(f)):
                    k = short(row["Kernel_Name"])
                    if not k.startswith("k_joint"):
                        continue
                    e = per.setdefault(k, {"dispatches": 0, "ms": 0.0})
                    e[row["Counter_Name"]] = e.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                    if row["Dispatch_Id"] not in seen:
                        seen.add(row["Dispatch_Id"])
                        e["dispatches"] += 1
                        e["ms"] += (float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) / 1e6
        else:
            for f in glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    k = short(row["Kernel_Name"])
                    if not k.startswith("k_joint"):
                        continue
                    e = per.setdefault(k, {"dispatches": 0, "ms": 0.0})
                    e["dispatches"] += 1
                    e["ms"] += (float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) / 1e6
        for e in per.values():       # the warm-up step is in there too: per dispatch
            e["ms_per_dispatch"] = e["ms"] / max(e["dispatches"], 1)
        return per
    finally:
        shutil.rmtree(out, ignore_errors=True)


GROUPS = [["TCP_UTCL1_REQUEST_sum", "TCP_UTCL1_TRANSLATION_MISS_sum", "TCP_UTCL1_TRANSLATION_HIT_sum", "TCP_PENDING_STALL_CYCLES_sum"],
          ["TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_STALL_sum", "TCC_EA0_WRREQ_64B_sum", "GRBM_GUI_ACTIVE"],
          ["TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum", "TCP_UTCL1_STALL_MULTI_MISS_sum", "TCP_UTCL1_STALL_INFLIGHT_MAX_sum", "TCP_UTCL1_THRASHING_STALL_sum"]]
t00 = time.time()
for n in sizes:
    e = res["runs"][str(n)] = {"reads": n}
    e["kept"] = plain(n)
    e["not_kept"] = plain(n, ["--joint-no-keep"])
    print(n, e, f"{time.time() - t00:.0f} s", flush=True)
    json.dump(res, open(out_path, "w"), indent=1)
for n in (sizes[0], sizes[-1]):
    e = res["runs"][str(n)]
    e["concurrent_per_kernel"] = profiled(n, None)
    print(n, "trace", f"{time.time() - t00:.0f} s", flush=True)
    e["serialised_counters"] = {}
    for g in GROUPS:
        got = profiled(n, g)
        if "error" in got:
            e["serialised_counters"].setdefault("errors", []).append(got["error"])
            continue
        for k, v in got.items():
            e["serialised_counters"].setdefault(k, {}).update(v)
        print(n, g[0], f"{time.time() - t00:.0f} s", flush=True)
    json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps(res)[:3000])

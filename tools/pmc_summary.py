"""Summarise rocprofv3 output directories into the files kept under profiles/.

    python tools/pmc_summary.py <tag> <stats_dir> <pmc_dir>... [--steps N] [--kernel SUBSTR]

<stats_dir>: a `rocprofv3 --kernel-trace --stats` run  -> profiles/<tag>_kernel_stats.csv (copied)
<pmc_dir>s : `rocprofv3 --pmc ... --kernel-trace` runs (one counter group each, N steps of bench.py)
             -> profiles/<tag>_pmc_counters.csv (per kernel name and counter: sum over dispatches)
                profiles/<tag>_pmc_traffic.json (per-step HBM bytes and VALU figures of the sweep kernels)
FETCH_SIZE / WRITE_SIZE are kilobytes (MI355X_MICROARCH.md, HBM section).  The guide's gfx950
correction (FETCH_SIZE x2) applies to 16-B-per-lane streaming reads; these kernels read dwords
and bytes, for which the guide calls the counter uncalibrated, so the raw value is reported and
the doubled one beside it as an upper bound.
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict


def find(d, suffix):
    return sorted(glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True))


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    opts = {a.split("=")[0]: a.split("=")[1] for a in sys.argv[1:] if a.startswith("--") and "=" in a}
    tag, stats_dir, pmc_dirs = args[0], args[1], args[2:]
    steps = int(opts.get("--steps", 1))
    kern = opts.get("--kernel", "k_sweep_")
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    st = find(stats_dir, "kernel_stats.csv")
    if st:
        shutil.copy(st[0], os.path.join(out_dir, f"{tag}_kernel_stats.csv"))
    total = defaultdict(float)          # (short kernel name, counter) -> sum
    ndisp = defaultdict(int)
    dur = defaultdict(float)
    for d in pmc_dirs:
        for f in find(d, "counter_collection.csv"):
            seen = set()
            for row in csv.DictReader(open(f)):
                name = row["Kernel_Name"].split("(")[0].replace("void ", "")
                total[(name, row["Counter_Name"])] += float(row["Counter_Value"])
                key = (row["Dispatch_Id"], row["Counter_Name"])
                if key not in seen and row["Counter_Name"]:
                    seen.add(key)
                ndisp[(name, row["Counter_Name"])] += 1
    with open(os.path.join(out_dir, f"{tag}_pmc_counters.csv"), "w") as fo:
        w = csv.writer(fo)
        w.writerow(["kernel", "counter", "sum_over_dispatches", "rows"])
        for (name, c), v in sorted(total.items()):
            w.writerow([name, c, v, ndisp[(name, c)]])
    # per kernel: duration of its dispatches in the counter passes (launches are serialised there) and the
    # shader clock GRBM_GUI_ACTIVE / 8 XCDs / duration (MI355X_MICROARCH.md, DVFS)
    per_kernel = {}
    for d in pmc_dirs:
        for f in find(d, "counter_collection.csv"):
            rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == "GRBM_GUI_ACTIVE"]
            for r in rows:
                name = r["Kernel_Name"].split("(")[0].replace("void ", "")
                if kern not in name:
                    continue
                e = per_kernel.setdefault(name, {"dispatches": 0, "ns": 0.0, "gui_active": 0.0})
                e["dispatches"] += 1
                e["ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                e["gui_active"] += float(r["Counter_Value"])
    for name, e in per_kernel.items():
        e["ms_per_dispatch_serialised"] = e["ns"] / e["dispatches"] / 1e6
        e["clock_GHz"] = e["gui_active"] / 8.0 / e["ns"]
        for c in ("SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"):
            if (name, c) in total:
                e[c + "_per_dispatch"] = total[(name, c)] / e["dispatches"]
        if e.get("SQ_INSTS_VALU_per_dispatch"):
            # SIMD-cycles the chip had per VALU wave-instruction while this kernel ran alone
            e["simd_cycles_per_valu_instruction"] = (e["gui_active"] / 8.0 / e["dispatches"]) * 1024.0 / e["SQ_INSTS_VALU_per_dispatch"]
    sweep = defaultdict(float)
    for (name, c), v in total.items():
        if kern in name:
            sweep[c] += v
    fetch, write = sweep.get("FETCH_SIZE", 0.0) * 1024 / steps, sweep.get("WRITE_SIZE", 0.0) * 1024 / steps
    valu = sweep.get("SQ_INSTS_VALU", 0.0) / steps
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import kernel_source_sha16
    res = {"command": "rocprofv3 --pmc <group> --kernel-trace --output-format csv -- python3 bench.py --steps "
                      f"{steps} --warmup 0 --cpu-sample 0 --one-shot-calls 0 --sub-configs none {opts.get('--flags', '')} "
                      "(one pass per counter group)",
           "source_sha16": kernel_source_sha16(),       # bench.py quotes these counters only for the same kernel sources
           "kernel": kern, "steps": steps,
           "hbm_bytes_per_step_sweep_kernels": fetch + write,
           "sweep_kernels": {"fetch_bytes": fetch, "fetch_bytes_x2_upper_bound": 2 * fetch, "write_bytes": write,
                             "valu_wave_instructions_per_step": valu,
                             "counters_per_step": {c: v / steps for c, v in sorted(sweep.items())}}}
    res["per_kernel_serialised"] = per_kernel
    ns = sum(e["ns"] for e in per_kernel.values())
    if ns > 0:      # shader clock over the sweep kernels' dispatches (serialised in the counter passes)
        res["clock_GHz"] = sum(e["gui_active"] for e in per_kernel.values()) / 8.0 / ns
    if sweep.get("SQ_ACTIVE_INST_VALU") and valu:
        # quad-cycles of VALU activity per wave instruction (MI355X_MICROARCH.md: SQ_ACTIVE_INST_* count quad-cycles)
        res["simd_cycles_per_valu_instruction_active"] = 4.0 * sweep["SQ_ACTIVE_INST_VALU"] / sweep["SQ_INSTS_VALU"]
    if sweep.get("SQ_WAVE_CYCLES") and sweep.get("SQ_ACTIVE_INST_VALU"):
        res["valu_active_fraction_of_wave_cycles"] = sweep["SQ_ACTIVE_INST_VALU"] / sweep["SQ_WAVE_CYCLES"]
        for c in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY"):
            if sweep.get(c):
                res[c + "_fraction_of_wave_cycles"] = sweep[c] / sweep["SQ_WAVE_CYCLES"]
    json.dump(res, open(os.path.join(out_dir, f"{tag}_pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()

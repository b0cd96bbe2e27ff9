"""Start / end of every kernel of the last step in a rocprofv3 --kernel-trace CSV: who overlaps whom, where the
device idles.  python3 tools/kernel_gantt.py <dir with *_kernel_trace.csv> [gap_us that separates steps = 1500]"""
import csv, glob, re, sys

root = sys.argv[1]
gap_us = float(sys.argv[2]) if len(sys.argv) > 2 else 1500.0
files = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Grid_Size_X", 0) or 0),
                     int(r.get("Workgroup_Size_X", 0) or 0), r.get("Queue_Id", "?"), int(r.get("VGPR_Count", 0) or 0)))
rows.sort()
# steps: split where the device idles for more than gap_us
steps, cur, last_end = [], [], None
for r in rows:
    if last_end is not None and r[0] - last_end > gap_us * 1e3 and cur:
        steps.append(cur); cur = []
    cur.append(r)
    last_end = r[1] if last_end is None else max(last_end, r[1])
if cur: steps.append(cur)
print(f"{len(rows)} dispatches, {len(steps)} busy stretches; the last complete ones:")
for st in steps[-3:-1] if len(steps) > 2 else steps[-1:]:
    t0 = st[0][0]
    busy_end = t0
    print(f"--- stretch of {len(st)} dispatches, {(max(r[1] for r in st) - t0) / 1e3:.1f} us")
    for s, e, name, grid, wg, q, vg in st:
        short = re.sub(r"\(.*", "", name).replace("void ", "")
        idle = max(0, s - busy_end) / 1e3
        print(f"{(s - t0) / 1e3:9.1f} -> {(e - t0) / 1e3:9.1f}  {(e - s) / 1e3:8.1f} us  waves {grid // 64:6d}  vgpr {vg:3d}  q{q}  {short}"
              + (f"   [device idle {idle:.1f} us before]" if idle > 5 else ""))
        busy_end = max(busy_end, e)

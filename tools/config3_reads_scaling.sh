#!/bin/bash
# bench.py --config 3 --reads N with the counters counted live, N = 1000 ... 80 000 (profiles/r04_config3_reads_scaling.json)
#   bash tools/config3_reads_scaling.sh [out dir]       (on the GPU box)
OUT=${1:-gpurun_out/c3scale}
mkdir -p $OUT
for n in 1000 2500 5000 20000 28000 40000 80000; do
  for keep in kept not_kept; do
    F=""; P="on"; if [ $keep = not_kept ]; then F="--joint-no-keep"; P="off"; fi
    timeout -k 10 400 python3 bench.py --config 3 --reads $n --steps 8 --warmup 2 --cpu-sample 0 --one-shot-calls 0 --sub-configs none --live-pmc $P $F > $OUT/${keep}_$n.json 2> $OUT/${keep}_$n.err || exit 1
  done
done
python3 - $OUT <<'P'
import json, sys
out = {"what": "bench.py --config 3 --reads N --steps 8 --warmup 2 on one MI355X (joint rounds 2+3, round 3 routed on the device): step time, device share, counted roofline (live rocprofv3 --pmc child passes), with the column states kept for round 3 and without (--joint-no-keep)", "runs": {}}
for n in (1000, 2500, 5000, 20000, 28000, 40000, 80000):
    e = out["runs"][str(n)] = {}
    for keep in ("kept", "not_kept"):
        d = json.loads(open(f"{sys.argv[1]}/{keep}_{n}.json").read().strip().splitlines()[-1])
        r = d["roofline"]
        e[keep] = {"ms_per_step": d["ms_per_step"], "value_Malign_per_s": d["value"] / 1e6, "device_ms_per_step": r.get("device_ms_per_step"),
                   "host_ms_per_step": r.get("host_ms_per_step"), "frac": r.get("frac"), "frac_over_issue_ceiling": r.get("frac_over_issue_ceiling"),
                   "frac_useful": r.get("frac_useful"), "traffic_bytes_per_step": r.get("traffic"), "us_per_read": 1e3 * d["ms_per_step"] / n}
json.dump(out, open(f"{sys.argv[1]}/summary.json", "w"), indent=1)
for n, e in out["runs"].items():
    print(n, {k: (round(v["ms_per_step"], 2), v["frac"]) for k, v in e.items()})
P

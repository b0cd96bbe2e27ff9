"""Samples the card's clocks, power and temperatures (sysfs; rocm-smi as a fallback) while a command runs.
python3 tools/dvfs_watch.py out.json -- <command ...>        (on the GPU box)"""
import glob, json, os, subprocess, sys, threading, time

out_path = sys.argv[1]
cmd = sys.argv[sys.argv.index("--") + 1:]


def cards():
    res = []
    for d in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
        if os.path.exists(os.path.join(d, "pp_dpm_sclk")) or glob.glob(os.path.join(d, "hwmon/hwmon*/power1_*")):
            res.append(d)
    return res


def read(path):
    try:
        return open(path).read().strip()
    except Exception:
        return None


def sample(d):
    s = {}
    sclk = read(os.path.join(d, "pp_dpm_sclk"))
    if sclk:
        cur = [ln for ln in sclk.splitlines() if ln.endswith("*")]
        s["sclk"] = cur[0] if cur else sclk.replace("\n", " | ")
    mclk = read(os.path.join(d, "pp_dpm_mclk"))
    if mclk:
        cur = [ln for ln in mclk.splitlines() if ln.endswith("*")]
        s["mclk"] = cur[0] if cur else None
    for h in glob.glob(os.path.join(d, "hwmon/hwmon*")):
        for name in ("power1_average", "power1_input", "temp1_input", "temp2_input", "temp3_input", "freq1_input", "freq2_input"):
            v = read(os.path.join(h, name))
            if v is not None:
                s[name] = int(v) if v.lstrip("-").isdigit() else v
    bp = read(os.path.join(d, "gpu_busy_percent"))
    if bp is not None:
        s["busy"] = bp
    return s


samples, stop = [], False
ds = cards()


def loop():
    while not stop:
        t = time.time()
        samples.append({"t": t, **{os.path.basename(os.path.dirname(d)): sample(d) for d in ds[:1]}})
        time.sleep(0.02)


th = threading.Thread(target=loop, daemon=True)
th.start()
t0 = time.time()
r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
stop = True
th.join(timeout=1)
smi = None
try:
    smi = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--showperflevel", "--json"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=30).stdout.decode()[-3000:]
except Exception as e:
    smi = repr(e)
json.dump({"cards": ds, "t0": t0, "command": cmd, "output": r.stdout.decode()[-6000:], "rc": r.returncode, "rocm_smi_after": smi,
           "samples": [{"t": round(s["t"] - t0, 3), **{k: v for k, v in s.items() if k != "t"}} for s in samples]}, open(out_path, "w"))
print(len(samples), "samples;", ds)
sys.exit(r.returncode)

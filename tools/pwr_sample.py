#!/usr/bin/env python3
"""Samples hwmon power1_input / freq1_input of every visible card while a command runs and reports the card that drew the most
power during it (a host shows all eight cards, other tenants' included: their sum says nothing):
    pwr_sample.py <label> -- <cmd...>
Prints the command's output, then one JSON line: mean / p10 / p90 of that card's package power (W) and shader clock (MHz) over the
samples taken while the command ran (first 15 % dropped as ramp).  bench.py's own telemetry (by PCI address) is the reference; this
is for C++ tools that do not use torch."""
import glob, json, os, subprocess, sys, threading, time


def rd(p):
    try:
        return float(open(p).read())
    except Exception:
        return float("nan")


label = sys.argv[1]
cmd = sys.argv[sys.argv.index("--") + 1:]
dirs = sorted(os.path.dirname(p) for p in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"))
samples, stop = [], False


def loop():
    while not stop:
        samples.append((time.time(), [rd(os.path.join(d, "power1_input")) * 1e-6 for d in dirs],
                        [rd(os.path.join(d, "freq1_input")) * 1e-6 for d in dirs]))
        time.sleep(0.0005)


t = threading.Thread(target=loop); t.start()
t0 = time.time()
r = subprocess.run(cmd, capture_output=True, text=True)
t1 = time.time()
stop = True; t.join()
sys.stdout.write(r.stdout)
s = [x for x in samples if x[0] >= t0 + 0.15 * (t1 - t0)]
if not s or not dirs:
    print(json.dumps({"label": label, "error": "no hwmon power files / no samples"}))
    sys.exit(0)
means = [sum(x[1][i] for x in s) / len(s) for i in range(len(dirs))]
best = max(range(len(dirs)), key=lambda i: means[i] if means[i] == means[i] else -1)


def st(k):
    v = sorted(x[k][best] for x in s)
    return {"mean": round(sum(v) / len(v), 1), "p10": round(v[len(v) // 10], 1), "p90": round(v[9 * len(v) // 10], 1)}


print(json.dumps({"label": label, "seconds": round(t1 - t0, 2), "samples": len(s), "card": dirs[best], "cards_seen": len(dirs),
                  "power_w": st(1), "sclk_mhz": st(2)}))

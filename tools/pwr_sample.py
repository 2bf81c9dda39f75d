#!/usr/bin/env python3
"""Samples the card's hwmon power1_input / freq1_input while a command runs:  pwr_sample.py <label> -- <cmd...>
Prints the command's output, then one JSON line: mean / p10 / p90 of package power (W) and shader clock (MHz) over the samples taken
while the command ran (first 15 % dropped as ramp)."""
import glob, json, subprocess, sys, threading, time

def rd(p):
    try:
        return float(open(p).read())
    except Exception:
        return float("nan")

label = sys.argv[1]
cmd = sys.argv[sys.argv.index("--") + 1:]
pw = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"))
fq = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input"))
samples, stop = [], False
def loop():
    while not stop:
        samples.append((time.time(), sum(rd(p) for p in pw) * 1e-6, max([rd(p) for p in fq] or [0]) * 1e-6))
        time.sleep(0.0005)
t = threading.Thread(target=loop); t.start()
t0 = time.time()
r = subprocess.run(cmd, capture_output=True, text=True)
t1 = time.time()
stop = True; t.join()
sys.stdout.write(r.stdout)
s = [x for x in samples if x[0] >= t0 + 0.15 * (t1 - t0)]
def st(i):
    v = sorted(x[i] for x in s)
    return {"mean": round(sum(v) / len(v), 1), "p10": round(v[len(v) // 10], 1), "p90": round(v[9 * len(v) // 10], 1)} if v else None
print(json.dumps({"label": label, "seconds": round(t1 - t0, 2), "samples": len(s), "power_w": st(1), "sclk_mhz": st(2)}))

#!/bin/bash
# which limiter holds the clock while the pipeline runs?  samples rocm-smi metrics beside a 20-step bench
cd /root/repo
O=gpurun_out/limiter; mkdir -p $O
(rocm-smi --showmetrics > $O/metrics_idle.txt 2>&1; true)
(rocm-smi -a > $O/smi_idle.txt 2>&1; true)
python bench.py --steps 60 --warmup 4 --no-cpu-baseline > $O/bench.json 2> $O/bench.err &
BP=$!
sleep 12
for i in 1 2 3 4 5 6; do
  (rocm-smi --showmetrics >> $O/metrics_load.txt 2>&1; true)
  (rocm-smi --showclocks --showpower --showtemp --showperflevel >> $O/smi_load.txt 2>&1; true)
  sleep 0.5
done
wait $BP
echo "bench rc $?"
grep -i -E "throttle|gfxclk|socket_power|temperature_hotspot|temperature_mem|curr_.*clk|accumulator|residency|limit" $O/metrics_load.txt | sort | uniq -c | sort -rn | head -60
tail -30 $O/smi_load.txt

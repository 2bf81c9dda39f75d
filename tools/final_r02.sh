# Final measurement batch of round 2 (run ON the GPU box; writes under gpurun_out/r02final, summaries are copied to profiles/ by hand)
set -u
O=gpurun_out/r02final
mkdir -p $O
export TMPDIR=/tmp
echo "== gpu tests"; python -m pytest tests -m gpu -q > $O/gputests.log 2>&1; echo "pytest rc=$?"; tail -3 $O/gputests.log
echo "== bench (default flags)"; python bench.py > $O/bench_final.json 2> $O/bench_final.err || { tail -5 $O/bench_final.err; exit 1; }
head -c 400 $O/bench_final.json; echo
echo "== bench --steps 20 --warmup 5 (the driver's invocation)"; python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2>> $O/bench_final.err || exit 1
echo "== kernel trace"; timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python bench.py --steps 3 --warmup 4 --no-extras --no-cpu-baseline --no-kernel-timer > $O/trace.log 2>&1; rc=$?; echo "trace rc=$rc"
if [ $rc -ge 124 ]; then exit $rc; fi
T=$(ls $O/trace/*/*kernel_trace.csv 2>/dev/null | head -1); if [ -n "$T" ]; then python tools/prof_summary.py $T --blocks 2 > $O/kernel_summary.md 2>$O/prof_summary.err; head -30 $O/kernel_summary.md; ls $O/trace/*/ ; fi
S=$(ls $O/trace/*/*kernel_stats.csv 2>/dev/null | head -1); if [ -n "$S" ]; then cp $S $O/kernel_stats.csv; fi
rm -f $O/trace/*/*kernel_trace.csv    # 60+ MB
echo "== pmc"; bash tools/collect_pmc.sh $O/pmc 5 2>&1 | tail -8
python tools/pmc_summary.py --work $O/pmc/kbench_shipped.log $O/pmc --json $O/pmc_shipped.json --md $O/pmc_shipped.md > /dev/null 2>$O/pmc_summary.err; cat $O/pmc_shipped.md | cut -c1-200
echo "== configs 3 / 4"; timeout -k 10 300 python tools/run_configs.py 240 > $O/configs34.json 2>$O/configs.err; echo "rc=$?"; head -c 600 $O/configs34.json; echo
echo "== config 5 (one replica, 960 latent frames, int8)"; timeout -k 10 300 python tools/run_configs.py 960 --quant int8 --only single > $O/config5.json 2>>$O/configs.err; echo "rc=$?"; head -c 500 $O/config5.json; echo

# Final measurement batch of round 3 (run ON the GPU box; writes under gpurun_out/r03final, summaries are copied to profiles/ by hand).
# Steps are joined so that a timed-out / killed GPU step stops the batch (no further GPU work after a hang).
set -u
O=gpurun_out/r03final
mkdir -p $O
export TMPDIR=/tmp
step() { echo "== $1"; }
dead() { [ $1 -eq 124 ] || [ $1 -eq 137 ]; }

step "smoke"; timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; rc=$?; echo "smoke rc=$rc"; tail -1 $O/smoke.log; dead $rc && exit $rc
step "gpu tests"; timeout -k 10 1100 python -m pytest tests -m gpu -q > $O/gputests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/gputests.log; dead $rc && exit $rc
[ $rc -eq 134 ] && exit $rc
step "bench --steps 20 --warmup 5 (the driver's invocation)"
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_final.json 2> $O/bench_final.err; rc=$?; echo "bench rc=$rc"; dead $rc && exit $rc
python3 - $O/bench_final.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(r["value"], r["unit"], "ms/step", round(r["ms_per_step"], 2), "roofline frac", round(r["roofline"]["frac"], 4), "avg_us", round(r["roofline"]["avg_us"], 1))
t = r.get("telemetry") or {}
print({k: (round(v, 1) if isinstance(v, float) else v) for k, v in t.items() if k not in ("gpu_metrics_delta", "source")})
print("cpu_baseline", r.get("cpu_baseline"))
for k in (r.get("extras") or {}).get("kernels", [])[:14]:
    print("  %-22s %-58s %8.1f us  frac %.3f" % (k.get("tag"), k.get("kernel", "")[:58], k.get("avg_us", 0), k.get("frac", 0)))
e = r.get("extras") or {}
print({k: e[k] for k in e if k != "kernels"})
PY
step "kernel trace of bench.py"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 3 --warmup 4 --no-extras --no-cpu-baseline --no-kernel-timer > $O/trace.log 2>&1; rc=$?; echo "trace rc=$rc"; dead $rc && exit $rc
T=$(ls $O/trace/*/*kernel_trace.csv 2>/dev/null | head -1); if [ -n "$T" ]; then python tools/prof_summary.py $T --blocks 2 > $O/kernel_summary.md 2>$O/prof_summary.err; head -24 $O/kernel_summary.md | cut -c1-200; fi
S=$(ls $O/trace/*/*kernel_stats.csv 2>/dev/null | head -1); if [ -n "$S" ]; then cp $S $O/kernel_stats.csv; fi
rm -f $O/trace/*/*kernel_trace.csv $O/trace/*/*.db
step "in-pipeline counters (layerseq)"
bash tools/pmc_inpipe.sh $O/pmc > $O/pmc.log 2>&1; rc=$?; tail -8 $O/pmc.log; dead $rc && exit $rc
python tools/pmc_inpipe_summary.py $O/pmc --md $O/pmc_inpipe.md --json $O/pmc_inpipe.json > /dev/null 2> $O/pmc_summary.err; grep "^|" $O/pmc_inpipe.md | cut -c1-220
find $O/pmc -name "*.db" -delete
step "energy per launch, generated vs HIP GEMMs"
for k in ffn1 ffn2 qkv o; do for t in 0 3; do echo "== kenergy $k gemm_asm=$t"; LL_TUNING=gemm_asm=$t timeout -k 10 60 ./tools/kenergy $k 0 2; done; done > $O/kenergy_gemm.txt 2>&1
echo "== kenergy ffn2 split-K (round 2's shipped form), gemm_asm=0" >> $O/kenergy_gemm.txt; KENERGY_SPLITK=1 LL_TUNING=gemm_asm=0 timeout -k 10 60 ./tools/kenergy ffn2 0 2 >> $O/kenergy_gemm.txt 2>&1
grep -h "kernel" $O/kenergy_gemm.txt | cut -c1-200
step "kbench gemmx, generated vs HIP"
for t in "gemm_asm=0" "gemm_asm=3"; do echo "== $t"; for shape in "4680 8960 1536 1" "4680 1536 8960 2" "4680 4608 1536 0" "4680 1536 1536 2" "4680 1536 1536 3" "4680 1536 1536 0"; do LL_TUNING=$t timeout -k 10 60 ./tools/kbench gemmx 20 $shape 2>&1 | grep -E "custom"; done; done | tee $O/kbench_gemm.txt
step "interleaved A/B of bench.py: gemm_asm=0 vs 3"
for i in 1 2; do for t in 0 3; do
  LL_TUNING=gemm_asm=$t timeout -k 10 400 python3 bench.py --steps 14 --warmup 4 --no-extras --no-cpu-baseline --no-kernel-timer > $O/ab_${t}_$i.json 2> $O/ab_${t}_$i.err; rc=$?; dead $rc && exit $rc
  python3 - $O/ab_${t}_$i.json $t <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
t = r.get("telemetry") or {}
g = t.get("gpu_metrics_delta") or {}
print(f"gemm_asm={sys.argv[2]}: {r['value']:.2f} {r['unit']}  ms/step {r['ms_per_step']:.1f}  sclk {t.get('sclk_mhz_avg', 0):.0f} MHz  power {t.get('power_w_avg', 0):.0f} W  ppt residency {g.get('ppt_residency_acc', 0) / max(1, g.get('accumulation_counter', 1)):.2f}")
PY
done; done | tee $O/ab_gemm_asm.txt
step "configs 3 / 4"; timeout -k 10 400 python tools/run_configs.py 240 > $O/configs34.json 2>$O/configs.err; rc=$?; echo "rc=$rc"; head -c 700 $O/configs34.json; echo; dead $rc && exit $rc
echo "done"

#!/bin/bash
# One parametrised script for everything that is run ON the GPU box (replaces round 3's per-call tools/runs/r03*.sh files):
#   bash tools/gpu_batch.sh <outdir-name> step [step ...]
# writes under gpurun_out/<outdir-name>/; summaries worth keeping are copied to profiles/ by hand.  Steps run in order and are
# joined so that a timed-out / killed GPU step stops the batch (no further GPU work after a hang).  Steps:
#   smoke | tests[=<pytest -k expr>] | testfile=<path>[::k] | tuned=<LL_TUNING_TEST spec> | bench | bench_short | trace | pmc_bench | pmc_inpipe
#   layerseq[=N] | layerexp | seqtrace=<env> | seqtracelib=<variant> | abenv=<ENV=..>[/rounds] | ablib=<variant>[/rounds] | abbench=<variant>[/rounds] | abbenchenv=<ENV=..>[/rounds] | ab=<LL_TUNING a>/<LL_TUNING b>[/rounds] | abseq=<LL_TUNING a>/<LL_TUNING b> | kbench=<args> | configs | nstreams[=<args>] | yardstick[=<args>]
set -u
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/$1; shift
mkdir -p $O
export TMPDIR=/tmp
dead() { [ $1 -eq 124 ] || [ $1 -eq 137 ] || [ $1 -eq 134 ] || [ $1 -eq 139 ]; }
benchline() {
python3 - "$1" <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("no bench line:", e); sys.exit(0)
t = r.get("telemetry") or {}
g = t.get("gpu_metrics_delta") or {}
rf = r.get("roofline") or {}
print(f"{r['value']:.2f} {r['unit']}  ms/step {r['ms_per_step']:.2f}  roofline frac {rf.get('frac', 0):.4f} avg_us {rf.get('avg_us', 0):.1f}  "
      f"sclk {t.get('sclk_mhz_avg', 0):.0f} MHz  power {t.get('power_w_avg', 0):.0f} W  ppt {g.get('ppt_residency_acc', 0) / max(1, g.get('accumulation_counter', 1)):.2f}")
PY
}
for step in "$@"; do
  name=${step%%=*}; arg=""; [ "$name" != "$step" ] && arg=${step#*=}
  echo "== $step"
  case $name in
    smoke)
      timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; rc=$?; echo "smoke rc=$rc"; tail -1 $O/smoke.log ;;
    tests)
      if [ -n "$arg" ]; then timeout -k 10 1100 python -m pytest tests -m gpu -q -x -k "$arg" > $O/gputests.log 2>&1; else timeout -k 10 1100 python -m pytest tests -m gpu -q > $O/gputests.log 2>&1; fi
      rc=$?; echo "pytest rc=$rc"; tail -5 $O/gputests.log; [ $rc -ne 0 ] && { cp $O/gputests.log $O/FAILED_gputests.log; grep -E "^(FAILED|ERROR)|Error|fault" $O/gputests.log | head -20; } ;;
    tuned)         # the model-level GPU suites under a kernel-family override: tuned=gemm_asm=0,attn_asm=0  (LL_TUNING_TEST)
      LL_TUNING_TEST="$arg" timeout -k 10 1100 python -m pytest tests/test_model_gpu.py tests/test_boundary_gpu.py tests/test_checkpoint.py -m gpu -q > $O/tuned_$(echo "$arg" | tr -c 'A-Za-z0-9\n' '_').log 2>&1
      rc=$?; echo "pytest rc=$rc"; tail -4 $O/tuned_$(echo "$arg" | tr -c 'A-Za-z0-9\n' '_').log ;;
    testfile)
      f=${arg%%::*}; k=""; [ "$f" != "$arg" ] && k=${arg#*::}
      if [ -n "$k" ]; then timeout -k 10 900 python -m pytest $f -m gpu -q -x -s -k "$k" > $O/test_$(basename $f .py).log 2>&1; else timeout -k 10 900 python -m pytest $f -m gpu -q -x -s > $O/test_$(basename $f .py).log 2>&1; fi
      rc=$?; echo "pytest rc=$rc"; tail -6 $O/test_$(basename $f .py).log ;;
    bench)
      timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; rc=$?; echo "bench rc=$rc"; benchline $O/bench.json ;;
    bench_short)
      timeout -k 10 400 python3 bench.py --steps 14 --warmup 4 --no-extras --no-cpu-baseline --no-kernel-timer > $O/bench_short.json 2> $O/bench_short.err; rc=$?; benchline $O/bench_short.json ;;
    trace)
      timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 3 --warmup 4 --no-extras --no-cpu-baseline --no-kernel-timer > $O/trace.log 2>&1; rc=$?; echo "trace rc=$rc"
      T=$(ls $O/trace/*/*kernel_trace.csv 2>/dev/null | head -1); [ -n "$T" ] && python tools/prof_summary.py $T --blocks 2 > $O/kernel_summary.md 2>$O/prof_summary.err && head -30 $O/kernel_summary.md | cut -c1-200
      S=$(ls $O/trace/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$S" ] && cp $S $O/kernel_stats.csv
      rm -f $O/trace/*/*kernel_trace.csv $O/trace/*/*.db ;;
    pmc_bench)      # ONE counter pass over bench.py itself, the program directly after `--`; if it dies the log is kept and nothing is retried
      timeout -k 10 420 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_bench -- python3 bench.py --steps 2 --warmup 4 --no-extras --no-cpu-baseline --no-kernel-timer > $O/pmc_bench.log 2>&1; rc=$?
      echo "pmc_bench rc=$rc"; tail -3 $O/pmc_bench.log | cut -c1-300
      C=$(ls $O/pmc_bench/*/*counter_collection.csv 2>/dev/null | head -1); [ -n "$C" ] && python tools/pmc_bench_summary.py $C > $O/pmc_bench.md 2> $O/pmc_bench_summary.err && cat $O/pmc_bench.md | cut -c1-200
      find $O/pmc_bench -name "*.db" -delete 2>/dev/null
      [ $rc -eq 139 ] && echo "rocprofv3 --pmc over bench.py crashed (log kept: $O/pmc_bench.log); batch stops" ;;
    pmc_inpipe)
      bash tools/pmc_inpipe.sh $O/pmc > $O/pmc.log 2>&1; rc=$?; tail -8 $O/pmc.log
      python tools/pmc_inpipe_summary.py $O/pmc --md $O/pmc_inpipe.md --json $O/pmc_inpipe.json > /dev/null 2> $O/pmc_summary.err; grep "^|" $O/pmc_inpipe.md | cut -c1-220
      find $O/pmc -name "*.db" -delete ;;
    layerseq)
      timeout -k 10 200 ./tools/kbench layerseq ${arg:-1500} 2>&1 | tee -a $O/layerseq.txt; rc=${PIPESTATUS[0]} ;;
    layerexp)      # timing-only: what removing a launch / balancing the attention grid could buy (tools/kbench.hip layerseq switches)
      rc=0
      for rep in 1 2; do
        for e in "" "KB_SKIP=7" "KB_SKIP=2" "KB_SKIP=5" "KB_SKIP=0,10" "KB_SKIP=8" "KB_ATTN_LQ=5376 KB_ATTN_S=17024" "KB_ATTN_LQ=5376 KB_ATTN_S=18752" "KB_SKIP=3"; do
          echo "-- [$e]"; env $e timeout -k 10 120 ./tools/kbench layerseq 900 2>&1 | grep -v "^$"; r=${PIPESTATUS[0]}; dead $r && { rc=$r; break 2; }
        done
      done > >(tee $O/layerexp.txt); wait ;;
    ab)
      a=$(echo "$arg" | cut -d/ -f1); b=$(echo "$arg" | cut -d/ -f2); n=$(echo "$arg" | cut -d/ -f3); n=${n:-2}; rc=0
      for i in $(seq 1 $n); do for t in "$a" "$b"; do
        LL_TUNING=$t timeout -k 10 400 python3 bench.py --steps 14 --warmup 4 --no-extras --no-cpu-baseline --no-kernel-timer > $O/ab_$i.json 2> $O/ab_$i.err; rc=$?; dead $rc && break 2
        echo -n "[$t] "; benchline $O/ab_$i.json
      done; done > >(tee -a $O/ab.txt); wait ;;
    abseq)
      a=$(echo "$arg" | cut -d/ -f1); b=$(echo "$arg" | cut -d/ -f2); rc=0
      for i in 1 2 3; do for t in "$a" "$b"; do echo -n "[$t] "; LL_TUNING=$t timeout -k 10 120 ./tools/kbench layerseq 1500; r=$?; dead $r && { rc=$r; break 2; }; done; done > >(tee -a $O/abseq.txt); wait ;;
    seqtrace)      # per-kernel durations of the layer sequence under timing-only switches: seqtrace="KB_SKIP=0,10"  (kernel trace of tools/kbench layerseq)
      tag=$(echo "$arg" | tr -c 'A-Za-z0-9\n' '_'); rc=0
      env $arg timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/seqtrace_$tag -- ./tools/kbench layerseq 300 > $O/seqtrace_$tag.log 2>&1; rc=$?
      grep layerseq $O/seqtrace_$tag.log | tail -1
      S=$(ls $O/seqtrace_$tag/*/*kernel_stats.csv 2>/dev/null | head -1)
      [ -n "$S" ] && python3 - "$S" <<'PY' | tee $O/seqtrace_$tag.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("  %-64.64s calls %5s avg %9.1f us  total %8.1f ms" % (r["Name"], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
      rm -f $O/seqtrace_$tag/*/*kernel_trace.csv $O/seqtrace_$tag/*/*.db ;;
    seqtracelib)   # the same under a variant library (tools/build_variant.sh): seqtracelib=<name>  -- timing-only decompositions (floor table)
      rc=0
      LD_LIBRARY_PATH=experiments/libs/$arg timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/seqtrace_$arg -- ./tools/kbench layerseq 300 > $O/seqtrace_$arg.log 2>&1; rc=$?
      grep layerseq $O/seqtrace_$arg.log | tail -1
      S=$(ls $O/seqtrace_$arg/*/*kernel_stats.csv 2>/dev/null | head -1)
      [ -n "$S" ] && python3 - "$S" <<'PY' | tee $O/seqtrace_$arg.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("  %-64.64s calls %5s avg %9.1f us  total %8.1f ms" % (r["Name"], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
      rm -f $O/seqtrace_$arg/*/*kernel_trace.csv $O/seqtrace_$arg/*/*.db ;;
    abenv)         # interleaved A/B of the layer sequence under kbench environment switches: abenv=<ENV=..>[/rounds]  (timing-only experiments)
      v=$(echo "$arg" | cut -d/ -f1); n=$(echo "$arg" | cut -d/ -f2); [ "$n" = "$v" ] && n=3; rc=0
      for i in $(seq 1 $n); do
        echo -n "[shipped] "; timeout -k 10 120 ./tools/kbench layerseq 1500; r=$?; dead $r && { rc=$r; break; }
        echo -n "[$v] "; env $v timeout -k 10 120 ./tools/kbench layerseq 1500; r=$?; dead $r && { rc=$r; break; }
      done > >(tee -a $O/abenv.txt); wait ;;
    ablib)         # interleaved A/B of the layer sequence: in-tree library vs a variant built by tools/build_variant.sh (ablib=<name>[/rounds])
      v=$(echo "$arg" | cut -d/ -f1); n=$(echo "$arg" | cut -d/ -f2); [ "$n" = "$v" ] && n=3; rc=0
      for i in $(seq 1 $n); do
        echo -n "[shipped] "; timeout -k 10 120 ./tools/kbench layerseq 1500; r=$?; dead $r && { rc=$r; break; }
        echo -n "[$v] "; LD_LIBRARY_PATH=experiments/libs/$v timeout -k 10 120 ./tools/kbench layerseq 1500; r=$?; dead $r && { rc=$r; break; }
      done > >(tee -a $O/ablib_$v.txt); wait ;;
    abbenchenv)    # interleaved A/B of bench.py under an environment switch: abbenchenv=<ENV=..>[/rounds]
      v=$(echo "$arg" | cut -d/ -f1); n=$(echo "$arg" | cut -d/ -f2); [ "$n" = "$v" ] && n=2; rc=0
      for i in $(seq 1 $n); do for t in "" "$v"; do
        env $t timeout -k 10 400 python3 bench.py --steps 14 --warmup 4 --no-extras --no-cpu-baseline --no-kernel-timer > $O/abe_$i.json 2> $O/abe_$i.err
        rc=$?; dead $rc && break 2
        echo -n "[${t:-shipped}] "; benchline $O/abe_$i.json
      done; done > >(tee -a $O/abbenchenv.txt); wait ;;
    abbench)       # the same with bench.py: abbench=<name>[/rounds]
      v=$(echo "$arg" | cut -d/ -f1); n=$(echo "$arg" | cut -d/ -f2); [ "$n" = "$v" ] && n=2; rc=0
      for i in $(seq 1 $n); do for t in "" "experiments/libs/$v/liblonglive_hip.so"; do
        if [ -z "$t" ]; then timeout -k 10 400 python3 bench.py --steps 14 --warmup 4 --no-extras --no-cpu-baseline --no-kernel-timer > $O/abb_$i.json 2> $O/abb_$i.err; else LONGLIVE_HIP_LIB=$t timeout -k 10 400 python3 bench.py --steps 14 --warmup 4 --no-extras --no-cpu-baseline --no-kernel-timer > $O/abb_$i.json 2> $O/abb_$i.err; fi
        rc=$?; dead $rc && break 2
        echo -n "[${t:-shipped}] "; benchline $O/abb_$i.json
      done; done > >(tee -a $O/abbench_$v.txt); wait ;;
    kbench)
      timeout -k 10 300 ./tools/kbench $arg 2>&1 | tee -a $O/kbench.txt; rc=${PIPESTATUS[0]} ;;
    configs)
      timeout -k 10 400 python tools/run_configs.py 240 > $O/configs34.json 2>$O/configs.err; rc=$?; echo "rc=$rc"; head -c 900 $O/configs34.json; echo ;;
    nstreams)      # nstreams[=<extra args>]: N = 1..4 interleaved prompt streams (tools/nstreams.py)
      timeout -k 10 600 python3 tools/nstreams.py $arg > $O/nstreams.json 2> $O/nstreams.err; rc=$?; echo "nstreams rc=$rc"; tail -c 700 $O/nstreams.json; echo ;;
    yardstick)     # same-device vendor yardstick (torch.mm / addmm / SDPA beside this library, tools/vendor_yardstick.py), then its kernel trace for the vendor kernels' names
      timeout -k 10 500 python3 tools/vendor_yardstick.py --out $O/yardstick.json ${arg} > /dev/null 2> $O/yardstick.txt; rc=$?; echo "yardstick rc=$rc"; cat $O/yardstick.txt | cut -c1-200
      if ! dead $rc; then
        timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ytrace -- python3 tools/vendor_yardstick.py --rounds 1 --iters 20 > /dev/null 2> $O/ytrace.log; r2=$?; echo "yardstick trace rc=$r2"
        S=$(ls $O/ytrace/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$S" ] && cp $S $O/yardstick_kernel_stats.csv && head -40 $S | cut -c1-220
        rm -f $O/ytrace/*/*kernel_trace.csv $O/ytrace/*/*.db; dead $r2 && rc=$r2
      fi ;;
    floor)         # floor=<variant>[,<variant>...]: the five block-linear launches looped alone (tools/kbench gemmx), the shipped library and
                   # timing-only variant builds (tools/build_variant.sh ... ASM_G_NO_*) interleaved, two rounds; 60 s limit per run; no profiler
      rc=0
      for rep in 1 2; do for v in shipped $(echo "$arg" | tr ',' ' '); do for shp in "4680 4608 1536 0" "4680 1536 1536 2" "4680 1536 1536 3" "4680 8960 1536 1" "4680 1536 8960 2"; do
        echo -n "[$v] "
        if [ "$v" = shipped ]; then timeout -k 5 60 ./tools/kbench gemmx 400 $shp; r=$?; else LD_LIBRARY_PATH=experiments/libs/$v timeout -k 5 60 ./tools/kbench gemmx 400 $shp; r=$?; fi
        [ $r -ne 0 ] && echo "(rc $r)"
        dead $r && { rc=$r; break 3; }
      done; done; done > >(tee -a $O/floor.txt); wait ;;
    benchq)        # benchq=<LL_TUNING a>/<LL_TUNING b>[/rounds]: INT8 (W8A8) pipeline, interleaved A/B WITH the per-kernel table of each (bench.py --quant int8)
      a=$(echo "$arg" | cut -d/ -f1); b=$(echo "$arg" | cut -d/ -f2); n=$(echo "$arg" | cut -d/ -f3); n=${n:-2}; rc=0
      for i in $(seq 1 $n); do for t in "$a" "$b"; do
        tag=$(echo "$t" | tr -c 'A-Za-z0-9\n' '_')
        LL_TUNING=$t timeout -k 10 400 python3 bench.py --quant int8 --steps 12 --warmup 4 --no-cpu-baseline --kernels-only > $O/benchq_${tag}_$i.json 2> $O/benchq_${tag}_$i.err; rc=$?; dead $rc && break 2
        echo -n "[int8 $t] "; benchline $O/benchq_${tag}_$i.json
      done; done > >(tee -a $O/benchq.txt); wait ;;
    clockmatrix)   # clockmatrix[=<variant>,<variant>...]: one kernel looped alone per load with clock / power / energy (tools/clock_matrix.py); the shipped
                   # library over every load, then each timing-only variant library over the generated GEMM loads
      rc=0
      timeout -k 10 200 python3 tools/clock_matrix.py > $O/clock_matrix.jsonl 2> $O/clock_matrix.err; rc=$?; echo "clockmatrix rc=$rc"
      if ! dead $rc; then for v in $(echo "$arg" | tr ',' ' '); do
        LONGLIVE_HIP_LIB=experiments/libs/$v/liblonglive_hip.so timeout -k 10 120 python3 tools/clock_matrix.py --loads ffn1,ffn2,qkv >> $O/clock_matrix.jsonl 2>> $O/clock_matrix.err; r=$?
        dead $r && { rc=$r; break; }
      done; fi
      python3 - $O/clock_matrix.jsonl <<'PY'
import json, sys
for line in open(sys.argv[1]):
    try: r = json.loads(line)
    except Exception: continue
    x = r.get("xcd_sclk_mhz_avg") or []
    print("%-12s %-12s %8.1f us %7s TF/s  sclk %6.0f MHz (xcd %s)  %6.0f W  ppt %.2f  %7.2f mJ/launch" % (r["lib"].split("/")[-2] if "/" in r["lib"] else r["lib"], r["load"], r["us_per_launch"], r["tflops"], r["sclk_mhz_avg"] or 0, ("%.0f-%.0f" % (min(x), max(x))) if x else "-", r["power_w_avg"] or 0, r["ppt_residency"] or 0, r["energy_mj_per_launch"]))
PY
      ;;
    vaeab)         # vaeab[=rounds]: VAE decoder alone (tools/vae_bench.py 9 2), fused conv + RMS_norm epilogues on / off (LL_VAE_FUSE), interleaved
      n=${arg:-2}; rc=0
      for i in $(seq 1 $n); do for t in 1 0; do
        echo -n "[LL_VAE_FUSE=$t] "; LL_VAE_FUSE=$t timeout -k 10 200 python3 tools/vae_bench.py 9 2 2>/dev/null | tail -1 | cut -c1-200; r=${PIPESTATUS[0]}; dead $r && { rc=$r; break 2; }
      done; done > >(tee -a $O/vaeab.txt); wait ;;
    clockloads)    # clockloads=<load>,<load>...: tools/clock_matrix.py over the named loads only (shipped library)
      timeout -k 10 200 python3 tools/clock_matrix.py --loads "$arg" > $O/clock_loads.jsonl 2> $O/clock_loads.err; rc=$?; echo "clockloads rc=$rc"; cut -c1-330 $O/clock_loads.jsonl ;;
    sweep)         # sweep=<key>:<v1>,<v2>,...[/rounds]: bench.py (short form) under LL_TUNING=<key>=<v>, every value in every round (interleaved)
      spec=$(echo "$arg" | cut -d/ -f1); n=$(echo "$arg" | cut -d/ -f2); [ "$n" = "$spec" ] && n=2; key=${spec%%:*}; vals=$(echo "${spec#*:}" | tr ',' ' '); rc=0
      for i in $(seq 1 $n); do for v in $vals; do
        LL_TUNING=$key=$v timeout -k 10 400 python3 bench.py --steps 14 --warmup 4 --no-extras --no-cpu-baseline --no-kernel-timer > $O/sweep_${key}_${v}_$i.json 2> $O/sweep_$i.err; rc=$?; dead $rc && break 2
        echo -n "[$key=$v] "; benchline $O/sweep_${key}_${v}_$i.json
      done; done > >(tee -a $O/sweep.txt); wait ;;
    stealprobe)    # synthetic upper bound of dynamic work stealing between XCDs (tools/xcd_balance_probe.hip, section "steal")
      timeout -k 10 300 ./tools/xcd_balance_probe steal > $O/stealprobe.txt 2>&1; rc=$?; echo "stealprobe rc=$rc"; cat $O/stealprobe.txt | cut -c1-260 ;;
    *) echo "unknown step $step"; rc=1 ;;
  esac
  if dead $rc; then echo "step $step died (rc $rc): batch stopped"; exit $rc; fi
done
echo "done"

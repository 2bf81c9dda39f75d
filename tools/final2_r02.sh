set -u
O=gpurun_out/r02final2
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
python -m pytest tests/test_ops_gpu.py -m gpu -q -k "v_insert or modulation or gemm" > $O/t.log 2>&1; echo "pytest rc=$?"; tail -1 $O/t.log
bash tools/collect_pmc.sh $O/pmc 5 2>&1 | tail -7
python tools/pmc_summary.py --work $O/pmc/kbench_shipped.log $O/pmc --json $O/pmc_shipped.json --md $O/pmc_shipped.md > /dev/null 2>$O/pmc_summary.err; cut -c1-210 $O/pmc_shipped.md

#!/bin/bash
# In-pipeline counters: rocprofv3 --pmc passes (the program directly after `--`, no tracing flags beside --pmc) over the model's launch
# sequence replayed by tools/kbench layerseq (no python under --pmc here), so that MFMA utilisation, wave-cycle shares and fabric traffic are stated for the regime the headline is measured in
# (kernels running back to back inside the 30-layer forward, the clock the pipeline holds), next to the tools/kbench figures.
# Run ON the GPU box:   bash tools/pmc_inpipe.sh <outdir>     then   python tools/pmc_inpipe_summary.py <outdir> --md ... --json ...
set -u
OUT=${1:-gpurun_out/pmc_inpipe}
mkdir -p "$OUT"
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
ARGS="layerseq 300"       # tools/kbench layerseq: one steady-state layer's 13 launches in model order, 300 layers = 2 AR blocks
pass() {
  name=$1; shift
  timeout -k 10 420 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- ./tools/kbench $ARGS > "$OUT/$name.log" 2>&1
  rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pass $name timed out / killed (rc $rc): stopping"; exit $rc; fi
  if [ $rc -ne 0 ]; then echo "pass $name failed (rc $rc), skipped"; tail -3 "$OUT/$name.log"; else echo "pass $name ok: $(grep layerseq "$OUT/$name.log" | head -1)"; fi
}
pass trace --kernel-trace
# (Round 3: a python / torch process under --pmc died in rocprofv3's dispatch hook at torch's int64 elementwise kernels -- synth's
# hash -- before any kernel of this library ran: gpurun_out/r03b/pmc/sq1.log.  No torch program is launched under --pmc from this
# script any more; bench.py's own pass is tools/pmc_bench.sh, taken once, after synth moved its hash into the library.)
pass sq1 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE
pass sq2 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
pass tcc_rd --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum GRBM_GUI_ACTIVE
pass tcc_wr --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum GRBM_GUI_ACTIVE
echo "done: $OUT"

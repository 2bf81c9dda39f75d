#!/bin/bash
# rocprofv3 counter passes over `tools/kbench shipped` (the launch set of one steady-state DiT layer), one pass per counter
# group (SQ: 8 slots, TCC: 4), no tracing flags beside --pmc.  Run ON the GPU box:   bash tools/collect_pmc.sh <outdir> [iters]
# then:  python tools/pmc_summary.py --work <outdir>/kbench_shipped.log <outdir> --json ... --md ...
# A pass that rocprofv3 rejects (unknown counter) is skipped; a pass that is killed or times out ends the script.
set -u
OUT=${1:-gpurun_out/pmc}
ITERS=${2:-5}
mkdir -p "$OUT"
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
./tools/kbench shipped 20 > "$OUT/kbench_shipped.log" 2>&1 || { echo "kbench failed"; tail -5 "$OUT/kbench_shipped.log"; exit 1; }
pass() {
  name=$1; shift
  timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- ./tools/kbench shipped "$ITERS" > "$OUT/$name.log" 2>&1
  rc=$?
  if [ $rc -ge 124 ]; then echo "pass $name killed (rc $rc): stopping"; exit $rc; fi
  if [ $rc -ne 0 ]; then echo "pass $name failed (rc $rc), skipped"; tail -3 "$OUT/$name.log"; else echo "pass $name ok"; fi
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
pass sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
pass sq3 SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE
pass tcc_rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum GRBM_GUI_ACTIVE
pass tcc_wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum GRBM_GUI_ACTIVE
pass tcc_hit TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
echo "done: $OUT"

set -u
O=gpurun_out/r03k; mkdir -p $O
./tools/kbench layerseq 300 | tee $O/layerseq.txt
bash tools/pmc_inpipe.sh $O/pmc; python tools/pmc_inpipe_summary.py $O/pmc --md $O/pmc_inpipe.md --json $O/pmc_inpipe.json > /dev/null 2> $O/pmc_summary.err; head -20 $O/pmc_inpipe.md | cut -c1-230
find $O/pmc -name "*.csv" -size +1M -delete
exit 0
for t in "attn_asm=1" "attn_asm=0"; do echo "== kenergy attn 2 ($t)"; LL_TUNING=$t ./tools/kenergy attn 2 4 2>&1 | tail -2; done | tee $O/kenergy_attn.txt

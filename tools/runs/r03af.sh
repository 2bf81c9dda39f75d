set -u
O=gpurun_out/r03af; mkdir -p $O
for t in "attn_pp_min_keys=1024" "attn_pp_min_keys=512" "attn_pp_min_keys=256"; do
  echo "== $t"; LL_TUNING=$t timeout -k 10 200 ./tools/kbench attn 20 2>&1 | sed -n '/attn_variant 2/,$p' | grep -E "cross|self-b1|long-1tile|ragged-long|FAIL"
done | tee $O/cross.txt

#!/bin/bash
# potential of FFN2 as 256x256 tiles x split-K 2 (emulated as N = 3072, K = 4480: 19 x 12 = 228 workgroups, no reduction)
cd /root/repo
for rep in 1 2; do
  ./tools/kbench gemmx 30 4680 1536 8960 0 | grep custom
  ./tools/kbench gemmx 30 4680 3072 4480 0 3 | grep custom
  ./tools/kbench gemmx 30 4680 3072 4480 0 4 | grep custom
  ./tools/kbench gemmx 30 4680 3072 4480 0 2 | grep custom
done

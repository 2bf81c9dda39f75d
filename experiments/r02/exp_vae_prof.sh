#!/bin/bash
# kernel trace of the VAE decoder bench
cd /root/repo
O=gpurun_out/vaeprof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /root/repo/$O -o vae -- python3 /root/repo/tools/vae_bench.py 9 2 > /root/repo/$O/run.log 2>&1 || { tail -5 /root/repo/$O/run.log; exit 1; }
cd /root/repo
ls $O | head
python tools/vae_prof_summary.py $O/vae_results.db > $O/summary.md 2>$O/summary.err || { cat $O/summary.err | tail -5; ls -R $O | head -20; exit 1; }
cat $O/summary.md | cut -c1-160

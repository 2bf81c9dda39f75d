mkdir -p gpurun_out/r02r
python -m pytest tests/test_model_gpu.py tests/test_shipped_sizes_gpu.py tests/test_cli_gpu.py -m gpu -x -q > gpurun_out/r02r/tests.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r02r/tests.log
for rep in 1 2; do for ov in 1 0; do
LL_OVERLAP=$ov timeout -k 10 200 python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-extras > gpurun_out/r02r/b_${ov}_$rep.json 2>/tmp/b.err || { tail -3 /tmp/b.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r02r/b_${ov}_$rep.json')); print('overlap=$ov', round(d['value'],2), 'f/s', round(d['ms_per_step'],2), 'attn', round(d['roofline']['avg_us'],1))"
done; done

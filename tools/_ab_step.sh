mkdir -p gpurun_out/r05_k
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-exact-fp32"
python -m pytest tests/test_gpu_ops.py -x -q -p no:cacheprovider > gpurun_out/r05_k/ops.log 2>&1; tail -n 1 gpurun_out/r05_k/ops.log
$B > gpurun_out/r05_k/b1_new.json 2> gpurun_out/r05_k/b1.err
CG_X3_GENERIC_EPILOGUE=1 $B > gpurun_out/r05_k/b2_generic.json 2> gpurun_out/r05_k/b2.err
$B > gpurun_out/r05_k/b3_new.json 2> gpurun_out/r05_k/b3.err
CG_X3_GENERIC_EPILOGUE=1 $B > gpurun_out/r05_k/b4_generic.json 2> gpurun_out/r05_k/b4.err
for f in gpurun_out/r05_k/b*.json; do python - $f <<'PY'
import json,sys
p=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=p['roofline']
print(sys.argv[1], p['value'], p['ms_per_step'], r['frac'], r['avg_launch_us'], r['conv_ms_per_step'], p['gpu_sensors']['sclk_mhz']['mean'])
PY
done

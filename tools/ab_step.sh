# same-call A/B of the whole step: $1 = output name, then pairs "label ENV=VALUE" (label new = no env)
OUT=gpurun_out/$1; shift
mkdir -p $OUT
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-exact-fp32 --no-live-pmc"
i=0
for spec in "$@" "$@"; do
  i=$((i+1)); label=${spec%%:*}; envs=${spec#*:}; [ "$envs" = "$spec" ] && envs=""
  env $envs $B > $OUT/b${i}_$label.json 2> $OUT/b${i}_$label.err
  python - $OUT/b${i}_$label.json <<'PY'
import json,sys
p=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=p['roofline']
print(sys.argv[1].split('/')[-1], p['value'], p['ms_per_step'], r['frac'], r['avg_launch_us'], r['conv_ms_per_step'], p['gpu_sensors']['sclk_mhz']['mean'])
PY
done

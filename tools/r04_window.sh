# the driver's 20-step window behind warm-ups of different length (fresh process each): what do the first steps of a run cost?
mkdir -p gpurun_out/r04
for w in 5 30 200 1000; do
  AZTOT_VERBOSE=1 timeout -k 10 300 python bench.py --steps 20 --warmup $w --no-cpu-baseline --no-steady > gpurun_out/r04/win_$w.json 2> gpurun_out/r04/win_$w.err; echo "warmup $w rc=$?"
  python - <<PY
import json
d=json.loads(open('gpurun_out/r04/win_$w.json').read().strip().splitlines()[-1])
print('warmup $w: ms/step', round(d['ms_per_step'],4), 'K', d['config'].get('sort_interval'))
PY
  grep -c "longest step" gpurun_out/r04/win_$w.err
done

# a workload under explicit skins: bash tools/r04_skin.sh WORKLOAD skin1 skin2 ...   (0 = the engine's own choice)
mkdir -p gpurun_out/r04
W=$1; shift
for s in "$@"; do
  AZTOT_VERBOSE=1 timeout -k 10 300 python bench.py --workload $W --skin $s --steps 300 --warmup 300 --no-cpu-baseline --no-steady > gpurun_out/r04/skin_${W}_$s.json 2> gpurun_out/r04/skin_${W}_$s.err; echo "$W skin $s rc=$?"
  python - <<PY
import json
d=json.loads(open('gpurun_out/r04/skin_${W}_$s.json').read().strip().splitlines()[-1])
k=d['kernels']
print('$W skin $s: ms/step', round(d['ms_per_step'],4), 'K', d['config'].get('sort_interval'), 'skin_A', round(d['config'].get('skin_A',0),3), 'cells', d['config'].get('n_cells'), {n:round(v['avg_us']) for n,v in k.items() if n in('pair_list','build_lists')}, 'T', round(d['config'].get('temperature_K',0),1))
PY
done

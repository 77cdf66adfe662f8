mkdir -p gpurun_out/r04
for n in oldfuse newfuse oldfuse newfuse; do for W in C2 C2T; do AZTOT_LIB=$PWD/ab/$n.so timeout -k 10 300 python bench.py --workload $W --steps 3000 --warmup 1000 --no-cpu-baseline --no-steady --no-profile > gpurun_out/r04/sm_${n}_$W.json 2>/dev/null; python -c "
import json;d=json.loads(open('gpurun_out/r04/sm_${n}_$W.json').read().strip().splitlines()[-1]);print('$n $W', round(d['ms_per_step']*1000,3),'us/step')"; done; AZTOT_LIB=$PWD/ab/$n.so timeout -k 10 300 python bench.py --case-study 1 --steps 3000 --warmup 1000 --no-steady --no-profile > gpurun_out/r04/sm_${n}_CS1.json 2>/dev/null; python -c "
import json;d=json.loads(open('gpurun_out/r04/sm_${n}_CS1.json').read().strip().splitlines()[-1]);print('$n CS1', round(d['ms_per_step']*1000,3),'us/step')"; done

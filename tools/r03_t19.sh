mkdir -p gpurun_out/r03
export HSA_ENABLE_IPC_MODE_LEGACY=0
for N in 2 4; do
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29800+N)) bench.py --gpus $N --steps 20 --warmup 5 --workload C2 > gpurun_out/r03/rehearse_$N.json 2> gpurun_out/r03/rehearse_$N.err
echo "N=$N rc=$?"; tail -3 gpurun_out/r03/rehearse_$N.err | cut -c1-300; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r03/rehearse_$N.json"))
    print(d["n_gpus"], d["ms_per_step"], d["config"]["transport"], d["config"]["sort_interval"], d["energy"]["engTot"])
except Exception as e: print("no json", e)
PY
done
python bench.py --workload C2 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('single', d['ms_per_step'], d['energy']['engTot'])"

mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "pair_lists or lazy or next_step or energies_only or any_cell or whole_number or golden_trajectory or initial_forces or thermalised or full_size" > gpurun_out/r03/t27.log 2>&1
echo rc=$?; tail -5 gpurun_out/r03/t27.log
for dbg in 0 16; do
python bench.py --workload C4 --steps 500 --warmup 500 --no-cpu-baseline --debug $dbg 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('C4 dbg $dbg', d['ms_per_step'], d['kernels']['pair_list'])"
python bench.py --workload C4T --steps 200 --warmup 200 --no-cpu-baseline --debug $dbg 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('C4T dbg $dbg', d['ms_per_step'], d['kernels']['pair_list'])"
done

mkdir -p gpurun_out/r03
timeout -k 10 1000 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_fullsize.py -k "not (initial_forces or potential_families or family_kernels or randomised or golden or cells_smaller)" > gpurun_out/r03/full3.log 2>&1
echo rc=$?; tail -15 gpurun_out/r03/full3.log

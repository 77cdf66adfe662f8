mkdir -p gpurun_out/r03
for s in 0.15 0.25 0 0.40 0.55; do
python bench.py --workload C4T --steps 300 --warmup 300 --no-cpu-baseline --skin $s 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); k=d['kernels']; print('C4T skin arg $s -> skin %.3f cells %d K %d: %.4f ms/step, pair_list %.1f build %.1f' % (d['config']['skin_A'], d['config']['n_cells'], d['config']['sort_interval'], d['ms_per_step'], k['pair_list']['avg_us'], k['build_lists']['avg_us']))"
done
for s in 0 0.40 -1; do
python bench.py --workload C4 --steps 500 --warmup 500 --no-cpu-baseline --skin $s 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); k=d['kernels']; print('C4  skin arg $s -> skin %.3f cells %d K %d: %.4f ms/step, pair_list %.1f' % (d['config']['skin_A'], d['config']['n_cells'], d['config']['sort_interval'], d['ms_per_step'], k['pair_list']['avg_us']))"
done

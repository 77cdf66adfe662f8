mkdir -p gpurun_out/r03
b() { name=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > gpurun_out/r03/j_$name.json 2> gpurun_out/r03/j_$name.err; echo "== $name rc=$?"; python - <<PY
import json
d=json.load(open("gpurun_out/r03/j_$name.json"))
print("%.4f raw %.4f K %s rebuilds %s" % (d["ms_per_step"], d["timed_window"]["ms_per_step_as_measured"], d["config"]["sort_interval"], d["timed_window"]["rebuilds_in_timed_region"]))
PY
}
for i in 1 2; do
b looks_d --workload C4
AZTOT_NO_LOOKS=1 b nolooks_d --workload C4
b looks_20 --workload C4 --steps 20 --warmup 5
AZTOT_NO_LOOKS=1 b nolooks_20 --workload C4 --steps 20 --warmup 5
done

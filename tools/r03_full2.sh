mkdir -p gpurun_out/r03
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r03/full4.log 2>&1
echo rc=$?; tail -6 gpurun_out/r03/full4.log
( time python bench.py > gpurun_out/r03/default_bench.json 2> gpurun_out/r03/default_bench.err ) 2>&1 | grep real
python - <<PY
import json
d=json.load(open("gpurun_out/r03/default_bench.json"))
print(d["value"], d["ms_per_step"], d["timed_window"], d.get("thermalised"))
PY
python -c "import __graft_entry__ as g; g.smoke()"

timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/t_all.log 2>&1; tail -3 gpurun_out/t_all.log
timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/bench17.json 2> gpurun_out/bench17.err
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/bench17.json").read().strip().splitlines()[-1])
print(d["value"], {k:(round(v.get("GB/s",v.get("TFLOP/s")))) for k,v in d["kernels"].items()})
PY

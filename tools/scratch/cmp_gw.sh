set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q > gpurun_out/t5.log 2>&1; tail -2 gpurun_out/t5.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_main.json 2> gpurun_out/bench_main.err
for v in A B C; do MAVA_LIB_PATH=tools/libmavahip_alt$v.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_alt$v.json 2> gpurun_out/bench_alt$v.err; done
python - <<PY
import json
for f in ("main","altA","altB","altC"):
    b=json.load(open(f"gpurun_out/bench_{f}.json")); print(f, round(b["value"]/1e6,2), round(b["ms_per_step"],3), b["kernel_ms_per_step"])
PY

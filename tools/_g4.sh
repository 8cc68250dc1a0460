cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_upfuse_gpu.py -x -q 2>&1 | tail -15
timeout 300 python tools/upfbench.py 2>&1 | tee gpurun_out/r04_upfbench.txt
timeout 200 python tools/upfbench.py --images 12 --groups 1 --raw 0 --rows 0,3,6 2>&1 | tee -a gpurun_out/r04_upfbench.txt
for f in 0 1; do
  FI_UPFUSE=$f timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-dice 2>gpurun_out/bench_upf$f.err | tail -1 > gpurun_out/bench_upf$f.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/bench_upf$f.json").read())
print("FI_UPFUSE=$f", d["value"], d["config"]["value_windows"], d["config"]["ms_per_aggregation_round"], d["roofline"]["min_roofline_frac"], d["roofline"]["kernel_time_breakdown_ms_per_step"])
PY
done

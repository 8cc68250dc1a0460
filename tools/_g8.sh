cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_unet_gpu.py tests/test_parity2_gpu.py tests/test_parity3_gpu.py tests/test_round4_gpu.py tests/test_losses_gpu.py -q 2>&1 | tail -6
timeout 100 python tools/narrowbench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_narrowbench.txt
timeout 100 python tools/upfbench.py --rows 0 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_upfbench_default.txt
name=ce
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-dice 2>gpurun_out/ab_$name.err | tail -1 > gpurun_out/ab_$name.json
python - <<PY
import json
d=json.loads(open("gpurun_out/ab_$name.json").read())
r=d["config"]["round_split_ms"]
print("$name", d["value"], d["config"]["value_windows"], "train", r["train"], "ala", r["ala"], "minroof", d["roofline"]["min_roofline_frac"], d["roofline"]["kernel_time_breakdown_ms_per_step"])
PY

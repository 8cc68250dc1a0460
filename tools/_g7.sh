cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_parity2_gpu.py -q -k "base_client_fit" 2>&1 | tail -4
timeout 300 python -m pytest tests/test_round4_gpu.py tests/test_upfuse_gpu.py tests/test_narrow_gpu.py -q 2>&1 | tail -4
ab() {  # name, env assignments...
  name=$1; shift
  env "$@" timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-dice 2>gpurun_out/ab_$name.err | tail -1 > gpurun_out/ab_$name.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab_$name.json").read())
r=d["config"]["round_split_ms"]
print("$name", d["value"], d["config"]["value_windows"], "train", r["train"], "ala", r["ala"], "minroof", d["roofline"]["min_roofline_frac"], "wgrad", d["roofline"]["kernel_time_breakdown_ms_per_step"].get("conv_wgrad"))
PY
}
ab base FI_DUMMY=1
ab rows64off FI_WGRAD_ROWS64=0
ab wgstream FI_WGRAD_STREAM=1
ab base2 FI_DUMMY=1

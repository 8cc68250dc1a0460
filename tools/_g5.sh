cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
for f in 0 1; do
  FEDICRA_PROBE_TAIL=$f timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-dice 2>gpurun_out/bench_tail$f.err | tail -1 > gpurun_out/bench_tail$f.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/bench_tail$f.json").read())
print("FEDICRA_PROBE_TAIL=$f", d["value"], d["config"]["value_windows"], d["config"]["round_split_ms"], d["roofline"]["min_roofline_frac"])
PY
done

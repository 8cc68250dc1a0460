cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_unet3d_gpu.py tests/test_unet3d_lc_gpu.py tests/test_extra_gpu.py tests/test_fullsize_gpu.py -q 2>&1 | tail -5
timeout 300 python bench.py --workload c4 --steps 40 --warmup 10 --no-cpu-baseline 2>gpurun_out/c4.err | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c4', d['value'], d['ms_per_step'], d['config'].get('hipgraph'), d['config'].get('hipgraph_error'), d['roofline'].get('kernel_time_breakdown_ms_per_step'))"

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r04_wgbench_rows64_wgs.txt; : > $O
for w in 256 320 448 640; do FI_WGRAD_ROWS64_WGS=$w timeout 120 python tools/wgbench2.py --min-c 32 2>&1 | grep -v amdgpu.ids | tee -a $O; done
FI_WGRAD_ROWS64_WGS=448 timeout 120 python tools/wgbench2.py --min-c 32 --full 1 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $O
FI_WGRAD_ROWS64_WGS=320 timeout 120 python tools/wgbench2.py --min-c 32 --full 1 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $O
timeout 200 python -m pytest tests/test_upfuse_gpu.py -q 2>&1 | tail -3

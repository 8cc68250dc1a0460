cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_round4_gpu.py -q -k "channel_rich" 2>&1 | tail -12
O=gpurun_out/r04_wgbench_rows64.txt; : > $O
timeout 120 python tools/wgbench2.py --rows 1 --min-c 32 2>&1 | grep -v amdgpu.ids | tee -a $O
timeout 120 python tools/wgbench2.py --rows 3 --min-c 32 2>&1 | grep -v amdgpu.ids | tee -a $O
for w in 160 512; do FI_WGRAD_ROWS64_WGS=$w timeout 120 python tools/wgbench2.py --rows 3 --min-c 32 2>&1 | grep -v amdgpu.ids | tee -a $O; done
for t in 42 24; do FI_WGRAD_ROWS64_TILE=$t timeout 120 python tools/wgbench2.py --rows 3 --min-c 32 2>&1 | grep -v amdgpu.ids | tee -a $O; done
timeout 120 python tools/wgbench2.py --rows 1 --min-c 32 --full 1 2>&1 | grep -v amdgpu.ids | tee -a $O
timeout 120 python tools/wgbench2.py --rows 3 --min-c 32 --full 1 2>&1 | grep -v amdgpu.ids | tee -a $O
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8

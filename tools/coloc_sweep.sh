run() { echo -n "$* : "; env "$@" timeout 200 python bench.py --steps 100 --warmup 20 --no-roofline --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
run A=1
run FI_TARGET_BLOCKS=512
run FI_TARGET_BLOCKS=2048
run FI_WGRAD_BLOCKS=256
run FI_WGRAD_BLOCKS=1024
run FI_TARGET_BLOCKS=512 FI_WGRAD_BLOCKS=256
run FI_PACK_BLOCKS=256
run FI_FWD_LDS_CAP=20000
run FI_FWD_LDS_CAP=80000

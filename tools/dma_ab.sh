for v in "" d1 d2 d8 d16 d3 d17 d26; do
  if [ -n "$v" ]; then export FEDICRA_HIP_LIB=$PWD/variants/dma_$v.so; else unset FEDICRA_HIP_LIB; fi
  echo "== variant ${v:-full}"
  timeout 120 python tools/kbench2.py --ws2 --only fused --layers 6,8,10 --cfgs dma 2>&1 | grep "dma:" | sed 's/v1.*|//'
done

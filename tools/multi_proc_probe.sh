#!/bin/bash
# How much aggregate throughput do several independent clients sharing ONE MI355X reach?  (probe for co-locating clients)
for n in 1 2 3 4; do
  echo "== $n concurrent process(es)"
  for i in $(seq 1 $n); do
    python bench.py --steps 300 --warmup 30 --no-roofline --no-cpu-baseline > /tmp/mp_$i.json 2>/dev/null &
  done
  wait
  for i in $(seq 1 $n); do python -c "import json,sys; d=json.load(open('/tmp/mp_$i.json')); print(d['value'], d['ms_per_step'])"; done
done

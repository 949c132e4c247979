#!/bin/bash
# same-box comparison of the masked step (config 5 shape) and the headline step under VGGP_DEEP_MIN_TILES settings
run() { timeout -k 10 200 python bench.py --masked --n 2048 --m 32 --steps 30 --warmup 5 --no-cpu 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('masked $1', round(d['ms_per_step'],4))"; }
runh() { timeout -k 10 200 python bench.py --no-cpu --no-extras --steps 400 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('headline $1', round(d['ms_per_step']*1e3,2))"; }
for rep in 1 2; do
  for t in 1024 256 128 64; do export VGGP_DEEP_MIN_TILES=$t; run $t; done
done
for rep in 1 2; do
  for t in 1024 128 64; do export VGGP_DEEP_MIN_TILES=$t; runh $t; done
done

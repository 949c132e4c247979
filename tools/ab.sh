#!/bin/bash
# usage: ab.sh ENVVAR  -- alternates default / ENVVAR=1 three times, prints ms per step
for rep in 1 2 3; do
  for v in 0 1; do
    if [ $v = 1 ]; then export $1=1; else unset $1; fi
    timeout -k 10 200 python bench.py --no-cpu --no-extras --steps 400 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$1' if $v else 'default', round(d['ms_per_step']*1e3,2))"
  done
done

#!/bin/bash
# eval-forward time per variants/librulgnn_*.so (development aid)
for lib in variants/librulgnn_*.so; do
  n=$(basename $lib .so | sed 's/librulgnn_//')
  RULGNN_LIB=$PWD/$lib python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); f=d['roofline_forward']
print('$n', 'eval fwd us', f['us_per_launch'], 'frac', f['frac'], 'step ms', d['ms_per_step'])"
done

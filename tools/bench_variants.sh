#!/bin/bash
# run bench.py once per variants/librulgnn_*.so and print step time + phase times (development aid)
for lib in variants/librulgnn_*.so; do
  n=$(basename $lib .so | sed 's/librulgnn_//')
  RULGNN_LIB=$PWD/$lib python bench.py --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['roofline']['phase_us']
print('$n', d['ms_per_step'], ' '.join(f'{k}={v}' for k,v in p.items()), 'sum=%.1f'%sum(p.values()))"
done

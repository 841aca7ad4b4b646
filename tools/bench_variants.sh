#!/bin/bash
# run bench.py once per variants/librulgnn_*.so (and per batch in $BATCHES) and print step time + phase times (development aid)
for lib in variants/librulgnn_*.so; do
  n=$(basename $lib .so | sed 's/librulgnn_//')
  for b in ${BATCHES:-65536}; do
  RULGNN_LIB=$PWD/$lib python bench.py --steps 30 --warmup 5 --batch $b --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['roofline']['phase_us']
print('$n', $b, d['ms_per_step'], '%.1f M/s' % (d['value']/1e6), ' '.join(f'{k}={v}' for k,v in p.items()), 'sum=%.1f'%sum(p.values()))"
  done
done

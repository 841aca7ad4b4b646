#!/bin/bash
# usage: tools/time_variants.sh "<command>"  -- runs the command once per variants/librulgnn_*.so (RULGNN_LIB override), development aid
for lib in variants/librulgnn_*.so; do
  n=$(basename $lib .so | sed 's/librulgnn_//')
  echo "== $n"; RULGNN_LIB=$PWD/$lib timeout 200 bash -c "$1" 2>&1 | tail -3
done

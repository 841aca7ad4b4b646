#!/bin/bash
for lib in variants/librulgnn_*.so; do
  n=$(basename $lib .so | sed 's/librulgnn_//')
  echo -n "$n "; RULGNN_LIB=$PWD/$lib python tools/time_stmsgcn.py 2>/dev/null | grep "bs128" | cut -c1-140
done

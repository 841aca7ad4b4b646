#!/bin/bash
# forward / backward time of the LSTM layer per variants/librulgnn_*.so (development aid)
for lib in variants/librulgnn_*.so; do
  n=$(basename $lib .so | sed 's/librulgnn_//')
  echo "== $n"; RULGNN_LIB=$PWD/$lib timeout 120 python tools/time_lstm.py 2>&1 | tail -3
done

#!/bin/bash
# eval time of HAGCN (LSTM-dominated) per variants/librulgnn_*.so (development aid)
for lib in variants/librulgnn_*.so; do
  n=$(basename $lib .so | sed 's/librulgnn_//')
  echo -n "$n "; RULGNN_LIB=$PWD/$lib python tools/time_hagcn.py 2>/dev/null | grep "bs256" | cut -c1-120
done

#!/bin/bash
# eval-forward kernel time per variants/librulgnn_*.so (development aid): tools/time_forward.py under each library
for lib in variants/librulgnn_*.so; do
  n=$(basename $lib .so | sed 's/librulgnn_//')
  echo "== $n"; RULGNN_LIB=$PWD/$lib python tools/time_forward.py "$@" 2>/dev/null | grep " mx "
done

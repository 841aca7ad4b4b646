#!/bin/bash
# development aid: per-kernel averages of the tiled XJTU step for a list of variant libraries (tools/build_variants.py)
# usage: tools/tiled_variants.sh <kernel-name-substring> base e1 e2 ...
cd /tmp && export TMPDIR=/tmp
pat=$1; shift
for v in "$@"; do
  lib=""; [ "$v" != base ] && lib="$GRAFT_REPO_ROOT/variants/librulgnn_$v.so"
  rm -rf /tmp/tv_$v
  (cd $GRAFT_REPO_ROOT && RULGNN_LIB=$lib rocprofv3 --kernel-trace --stats -d /tmp/tv_$v -o p --output-format csv -- python tools/time_tiled_one.py > /dev/null 2>&1)
  f=$(find /tmp/tv_$v -name '*kernel_stats.csv' | head -1)
  echo "== $v"; python3 - "$f" "$pat" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)/12e3
print(f"  step kernel time {tot:8.1f} us")
for r in rows:
    if any(p in r['Name'] for p in sys.argv[2].split(',')): print(f"  {r['Name'][:60]:60s} calls={r['Calls']:>4} avg_us={float(r['AverageNs'])/1e3:8.1f}")
PY
done

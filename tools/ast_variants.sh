#!/bin/bash
# development aid: per-kernel averages of the ASTGCNN step for variant libraries (tools/build_variants.py astgcnn.hip ...)
cd /tmp && export TMPDIR=/tmp
pat=$1; shift
for v in "$@"; do
  lib=""; [ "$v" != base ] && lib="$GRAFT_REPO_ROOT/variants/librulgnn_$v.so"
  rm -rf /tmp/av_$v
  (cd $GRAFT_REPO_ROOT && RULGNN_LIB=$lib rocprofv3 --kernel-trace --stats -d /tmp/av_$v -o p --output-format csv -- python bench.py --family ASTGCNN --steps 30 --warmup 5 --no-roofline --no-cpu-baseline > /dev/null 2>&1)
  f=$(find /tmp/av_$v -name '*kernel_stats.csv' | head -1)
  echo "== $v"; python3 - "$f" "$pat" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if any(p in r['Name'] for p in sys.argv[2].split(',')): print(f"  {r['Name'][:70]:70s} calls={r['Calls']:>4} avg_us={float(r['AverageNs'])/1e3:8.1f}")
PY
done

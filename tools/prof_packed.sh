#!/bin/bash
# kernel durations of the packed-row variants (rocprofv3 --kernel-trace --stats), one process each
set -e
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_packed
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in "bytes 0" "packed_obs 0" "packed 0" "packed_nogebv 0" "unpack 0" "pgebv 0" "bytes 1" "packed_obs 1" "packed 1"; do
  set -- $v
  d=$OUT/$1_$2
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python3 $REPO/tools/prof_packed.py $1 $2 > $d.log 2>&1
  f=$(find $d -name '*kernel_stats.csv' | head -1)
  echo "== $1 mode=$2" >> $OUT/summary.txt
  python3 - "$f" >> $OUT/summary.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f"  {r['Name'][:70]:70s} calls={r['Calls']:>5s} avg={float(r['AverageNs'])/1e3:8.2f} us")
PY
done
cat $OUT/summary.txt

#!/bin/bash
# Run ON THE GPU BOX: A/B of libraries built HERE beforehand (tools/ab_build.sh <name> ... -> tools/_ab/<name>/libbreedhip.so),
# each put in the shipped library's place in the box's scratch copy and run under rocprofv3's kernel trace -- no rebuild on
# the box's clock.  "base" = the shipped library.  Prints, per variant and round, the bench line's timings and the average
# duration of every kernel of the run.
#   usage: ab_prebuilt.sh out.txt c3|c5|c4|headline base w8 ...        (ROUNDS=2: every variant twice, interleaved)
cd "$(dirname "$0")/.."
R=$(pwd)
OUT=$R/$1; CFG=$2; shift 2
: > $OUT
cp breedgym_amd/libbreedhip.so /tmp/libbreedhip_shipped.so
if [ $CFG = headline ]; then ARGS="--headline-only --no-cpu-baseline --no-live-traffic --steps 400 --warmup 20 --long-steps 0"; else ARGS="--only $CFG --steps ${STEPS:-200} --warmup 20"; fi
cd /tmp && export TMPDIR=/tmp
for round in $(seq 1 ${ROUNDS:-2}); do
for name in "$@"; do
  if [ $name = base ]; then cp /tmp/libbreedhip_shipped.so $R/breedgym_amd/libbreedhip.so; else cp $R/tools/_ab/$name/libbreedhip.so $R/breedgym_amd/libbreedhip.so; fi
  echo "=== $name (round $round)" >> $OUT
  rm -rf /tmp/abp; mkdir -p /tmp/abp
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abp -o p -- python3 $R/bench.py $ARGS > /tmp/abp/line.json 2> /tmp/abp/err.log || { echo FAILED >> $OUT; tail -5 /tmp/abp/err.log >> $OUT; }
  python3 - >> $OUT 2>&1 <<'PY'
import csv, glob, json
try:
    d = json.loads(open("/tmp/abp/line.json").read().strip().splitlines()[-1])
    def walk(p, x):
        if isinstance(x, dict):
            for k, v in x.items(): walk(p + "." + k, v)
        elif isinstance(x, (int, float)) and p.split(".")[-1] in ("ms_per_step", "launch_ms", "kernel_ms", "ms_per_call", "ms_per_generation"): print("  ", p, x)
    walk("", d)
except Exception as e:
    print("   no line:", e)
for f in glob.glob("/tmp/abp/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if float(row["Percentage"]) >= 0.3:
            print("   %-70s calls %5s avg %9.2f us  min %9.2f" % (row["Name"][:70], row["Calls"], float(row["AverageNs"]) / 1e3, float(row["MinNs"]) / 1e3))
PY
done
done
cp /tmp/libbreedhip_shipped.so $R/breedgym_amd/libbreedhip.so
cat $OUT

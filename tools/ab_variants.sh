#!/bin/bash
# Run ON THE GPU BOX: rebuild libbreedhip.so with each set of -D flags and time the headline launch with it.
#   usage: ab_variants.sh out.txt "" "-DBH_EXP_X" ...     (one box, one run: the variants are comparable)
cd "$(dirname "$0")/.."
OUT=$1; shift
: > $OUT
for round in $(seq 1 ${ROUNDS:-2}); do
for FLAGS in "$@"; do
  python3 - "$FLAGS" <<'PY' >> $OUT 2>&1
import sys
sys.path.insert(0, ".")
from breedgym_amd import _lib
_lib.build_native(force=True, extra_flags=tuple(sys.argv[1].split()))
PY
  echo "=== flags: [$FLAGS]" >> $OUT
  if [ "${MODE:-headline}" = tool ]; then
  python3 $TOOL >> $OUT 2>&1
  elif [ "${MODE:-headline}" = headline ]; then
  python3 bench.py --headline-only --no-cpu-baseline --no-live-traffic --steps 400 --warmup 20 --long-steps 2000 2>>$OUT | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('value %.1f M  ms_per_step %.5f  long %.5f  kernel_us %s  frac %.3f' % (d['value']/1e6, d['ms_per_step'], d.get('ms_per_step_long',0), r.get('kernel_us', r.get('kernel_ms')), r['frac']))" >> $OUT
  else
  python3 bench.py --only $MODE --steps 300 --warmup 30 2>>$OUT | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
def walk(p,x):
    if isinstance(x,dict):
        for k,v in x.items(): walk(p+'.'+k,v)
    elif isinstance(x,(int,float)) and ('ms' in p.split('.')[-1] or 'us' in p.split('.')[-1]): print(p, x)
walk('',d)" >> $OUT
  fi
done
done
python3 - <<'PY' >> $OUT 2>&1
import sys
sys.path.insert(0, ".")
from breedgym_amd import _lib
_lib.build_native(force=True)
PY
cat $OUT

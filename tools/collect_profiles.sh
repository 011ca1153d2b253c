#!/bin/bash
# Run ON THE GPU BOX (via gpurun): the rocprofv3 passes behind profiles/<round>_*.   usage: collect_profiles.sh r03
#   kernel-trace --stats of the headline configuration alone, of configs[2], configs[3] (one rank) and configs[4] alone;
#   separate --pmc passes (FETCH_SIZE / WRITE_SIZE / SQ) of the headline and of configs[2], FETCH/WRITE of configs[4].
# Distil with tools/distil_profiles.py <round> afterwards.
set -e
cd "$(dirname "$0")/.."
R=$(pwd)
TAG=${1:-r05}
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
HL="python3 $R/bench.py --steps 100 --warmup 10 --headline-only"
C3="python3 $R/bench.py --steps 100 --warmup 10 --only c3"
C3L="python3 $R/bench.py --steps 100 --warmup 10 --only c3lazy"
C5="python3 $R/bench.py --steps 100 --warmup 10 --only c5"
C4="python3 $R/bench.py --steps 100 --warmup 10 --only c4"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_c2 -o p -- $HL > $OUT/kt_c2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_c3 -o p -- $C3 > $OUT/kt_c3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_c5 -o p -- $C5 > $OUT/kt_c5.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_c3lazy -o p -- $C3L > $OUT/kt_c3lazy.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_c4 -o p -- $C4 > $OUT/kt_c4.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_wheat -o p -- python3 $R/tools/prof_wheat.py 40 > $OUT/kt_wheat.log 2>&1
for cfg in c2 c3; do
  if [ $cfg = c2 ]; then CMD=$HL; else CMD=$C3; fi
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$cfg -o p -- $CMD > $OUT/fetch_$cfg.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write_$cfg -o p -- $CMD > $OUT/write_$cfg.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_WR \
    --output-format csv -d $OUT/sq_$cfg -o p -- $CMD > $OUT/sq_$cfg.log 2>&1
done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_c5 -o p -- $C5 > $OUT/fetch_c5.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write_c5 -o p -- $C5 > $OUT/write_c5.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_WR \
  --output-format csv -d $OUT/sq_c5 -o p -- $C5 > $OUT/sq_c5.log 2>&1
echo done; find $OUT -name "*.csv" | wc -l

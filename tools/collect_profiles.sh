#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 passes over the bench command.
#   pass 1: --kernel-trace --stats            -> per-kernel durations
#   pass 2: --pmc FETCH_SIZE                  -> bytes read through the L2's memory side
#   pass 3: --pmc WRITE_SIZE                  -> bytes written
#   pass 4: SQ counters of the hot kernel
# Output under gpurun_out/prof_$1/ ; distil with tools/distil_profiles.py afterwards.
set -e
TAG=${1:-r01}
cd "$(dirname "$0")/.."
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
CMD="python3 bench.py --steps 50 --warmup 10 --headline-only"
ALL="python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $CMD > $OUT/kt.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_all -- $ALL > $OUT/kt_all.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU \
  --output-format csv -d $OUT/sq -- $CMD > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU \
  --output-format csv -d $OUT/sq_all -- $ALL > $OUT/sq_all.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum \
  --output-format csv -d $OUT/tcc -- $CMD > $OUT/tcc.log 2>&1 || true
echo done; find $OUT -name "*.csv" | head -30

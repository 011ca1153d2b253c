#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 --kernel-trace of the headline stepping loop, then per-dispatch durations by kernel and
# grid size in the steady state + a stretch of the timeline (tools/trace_analyze.py).  profiles/r04_fused_reset.txt.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/kt_steps; mkdir -p $R/gpurun_out/kt_steps
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kt_steps -o p -- python3 $R/bench.py --headline-only --no-cpu-baseline --no-live-traffic --steps 200 --warmup 20 --long-steps 200 > $R/gpurun_out/kt_steps/log.txt 2>&1
python3 $R/tools/trace_analyze.py $R/gpurun_out/kt_steps/p_kernel_trace.csv

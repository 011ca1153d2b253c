#!/bin/bash
# TIMING-ONLY variants of the library behind profiles/r05_probe_overlap.txt sections 4-6: what would the headline launch
# gain if consecutive generations' launches could overlap?  Built HERE by tools/ab_build.sh (copies of csrc with sed
# edits, into tools/_ab/<name>/ with tools/probe_overlap.cpp as the driver); they give WRONG offspring (races on the
# two-block rings, no fences) and exist to be timed:
#   twostream  the two-role launch alternates over two non-blocking HIP streams, nothing orders generation g+1 behind g
#   depwait    + every row wave adds 1 to ONE device counter at its end, and waits in front of its first parent words
#              for the counter to reach all row waves of the launches before (same-address atomics: the way NOT to do it)
#   depflags   + every row wave stores the launch number into its own flag word at its end; wave 0 of a workgroup
#              polls the 128 flag words of its env (one 512-byte read) in front of the first parent words, the other
#              row waves wait for it in LDS
#   depfence   + agent-scope release / acquire fences around the flags (at the end of this file; depfenceR / depfenceA:
#              only the one or the other, built by hand the same way) -- 175 / 164 / 30 us: fences per wave are out
# run on the GPU box:   tools/_ab/<name>/driver 3000 1 16     (generations, only the one-launch arrangement, envs)
set -e
cd "$(dirname "$0")/.."
LAUNCH='hipLaunchKernelGGL\(kernel, dim3\(\(unsigned\)grid\), dim3\(512\), lds, c->stream, rp, mp, fp, gp\);'
STREAMS='s|#include "ctx.h"|#include "ctx.h"\nstatic hipStream_t ab_stream() { static hipStream_t s[2]; static int i = -1; if (i < 0) { hipStreamCreateWithFlags(\&s[0], hipStreamNonBlocking); hipStreamCreateWithFlags(\&s[1], hipStreamNonBlocking); i = 0; } i ^= 1; return s[i]; }|'
DONE_AT='^(        if \(lane == 0 \&\& p.reward\) reward_wave_done\(rw, p.reward, p.reward_pitch, env0, expected\);)$'
LDS_WAIT='else { for (int spin = 0; spin < (1 << 20); ++spin) { if (__hip_atomic_load(\&ab_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == p.ab_expect + 1u) break; __builtin_amdgcn_s_sleep(2); } }'
LDS_GO='if (lane == 0) __hip_atomic_store(\&ab_flag, p.ab_expect + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);'

tools/ab_build.sh twostream --driver tools/probe_overlap.cpp \
  "s|$LAUNCH|hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(512), lds, ab_stream(), rp, mp, fp, gp);|" "$STREAMS"

tools/ab_build.sh depwait --driver tools/probe_overlap.cpp \
  "s|$LAUNCH|static unsigned ab_total = 0; rp.ab_expect = ab_total; ab_total += 4u * (unsigned)rp.n_blocks; hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(512), lds, ab_stream(), rp, mp, fp, gp);|" "$STREAMS" \
  's|^struct PackedParams \{|static __device__ unsigned ab_done;\nstruct PackedParams {\n  unsigned ab_expect;|' \
  "s|^      Words wa, wb;\$|      if constexpr (MODE == MODE_CROSS \&\& DIGLDS) { __shared__ unsigned ab_flag; if (wave == 0) { for (int spin = 0; spin < (1 << 12); ++spin) { if (__hip_atomic_load(\&ab_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= p.ab_expect) break; __builtin_amdgcn_s_sleep(16); } $LDS_GO } $LDS_WAIT }\n      Words wa, wb;|" \
  "s|$DONE_AT|\1\n    if constexpr (MODE == MODE_CROSS \&\& DIGLDS) if (lane == 0) __hip_atomic_fetch_add(\&ab_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);|"

tools/ab_build.sh depflags --driver tools/probe_overlap.cpp \
  "s|$LAUNCH|static unsigned ab_launch = 0; rp.ab_expect = ab_launch++; hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(512), lds, ab_stream(), rp, mp, fp, gp);|" "$STREAMS" \
  's|^struct PackedParams \{|static __device__ unsigned ab_flags[8192];\nstruct PackedParams {\n  unsigned ab_expect;|' \
  "s|^      Words wa, wb;\$|      if constexpr (MODE == MODE_CROSS \&\& DIGLDS) { __shared__ unsigned ab_flag; if (wave == 0) { const unsigned *f = ab_flags + (unsigned)(block / 32) * 128u; for (int spin = 0; spin < (1 << 12); ++spin) { const unsigned a = __hip_atomic_load(f + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), b = __hip_atomic_load(f + 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (__all(a >= p.ab_expect \&\& b >= p.ab_expect)) break; __builtin_amdgcn_s_sleep(8); } $LDS_GO } $LDS_WAIT }\n      Words wa, wb;|" \
  "s|$DONE_AT|\1\n    if constexpr (MODE == MODE_CROSS \&\& DIGLDS) if (lane == 0) __hip_atomic_store(ab_flags + (unsigned)block * 4u + (unsigned)wave, p.ab_expect + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);|"

# depfence = depflags + what the memory model asks for around the flags: an agent-scope RELEASE fence in front of a row
# wave's flag store (its rows -- plain stores for the state, write-through for the observation -- visible to the other
# XCDs) and an agent-scope ACQUIRE fence behind the wait of every row wave (the parents' rows not out of this XCD's L2)
tools/ab_build.sh depfence --driver tools/probe_overlap.cpp \
  "s|$LAUNCH|static unsigned ab_launch = 0; rp.ab_expect = ab_launch++; hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(512), lds, ab_stream(), rp, mp, fp, gp);|" "$STREAMS" \
  's|^struct PackedParams \{|static __device__ unsigned ab_flags[8192];\nstruct PackedParams {\n  unsigned ab_expect;|' \
  "s|^      Words wa, wb;\$|      if constexpr (MODE == MODE_CROSS \&\& DIGLDS) { __shared__ unsigned ab_flag; if (wave == 0) { const unsigned *f = ab_flags + (unsigned)(block / 32) * 128u; for (int spin = 0; spin < (1 << 12); ++spin) { const unsigned a = __hip_atomic_load(f + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), b = __hip_atomic_load(f + 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (__all(a >= p.ab_expect \&\& b >= p.ab_expect)) break; __builtin_amdgcn_s_sleep(8); } $LDS_GO } $LDS_WAIT __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, \"agent\"); }\n      Words wa, wb;|" \
  "s|$DONE_AT|\1\n    if constexpr (MODE == MODE_CROSS \&\& DIGLDS) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, \"agent\"); if (lane == 0) __hip_atomic_store(ab_flags + (unsigned)block * 4u + (unsigned)wave, p.ab_expect + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }|"

# depsc1 = depflags + coherence on the rows themselves instead of fences: the state rows stored as relaxed agent-scope
# atomics (global_store_short sc1: write-through) with s_waitcnt vmcnt(0) in front of the wave's flag store (the flag
# only once the rows are out), the parents' words loaded as relaxed agent-scope atomics (global_load_ushort sc1: past
# this XCD's L2; the mask words stay plain loads -- masks made two generations ahead would have a kernel boundary
# between maker and reader).  depsc1S: only the stores + the wait; depsc1L: only the loads.
WAITHEAD="      if constexpr (MODE == MODE_CROSS \&\& DIGLDS) { __shared__ unsigned ab_flag; if (wave == 0) { const unsigned *f = ab_flags + (unsigned)(block / 32) * 128u; for (int spin = 0; spin < (1 << 12); ++spin) { const unsigned a = __hip_atomic_load(f + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), b = __hip_atomic_load(f + 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (__all(a >= p.ab_expect \&\& b >= p.ab_expect)) break; __builtin_amdgcn_s_sleep(8); } $LDS_GO } $LDS_WAIT"
FLAG="if (lane == 0) __hip_atomic_store(ab_flags + (unsigned)block * 4u + (unsigned)wave, p.ab_expect + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);"
SC1LOAD='__hip_atomic_load(reinterpret_cast<const uint16_t *>(reinterpret_cast<const char *>(\1) + 2u * (uint32_t)v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)'
SC1STORE='s|^          \*reinterpret_cast<uint16_t \*>\(reinterpret_cast<char \*>\(dstp\[r\]\) \+ 2u \* \(uint32_t\)vu\) = \(uint16_t\)o;|          __hip_atomic_store(reinterpret_cast<uint16_t *>(reinterpret_cast<char *>(dstp[r]) + 2u * (uint32_t)vu), (uint16_t)o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);|'
for v in depsc1 depsc1S depsc1L; do
  edits=("s|$LAUNCH|static unsigned ab_launch = 0; rp.ab_expect = ab_launch++; hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(512), lds, ab_stream(), rp, mp, fp, gp);|" "$STREAMS"
         's|^struct PackedParams \{|static __device__ unsigned ab_flags[8192];\nstruct PackedParams {\n  unsigned ab_expect;|'
         "s|^      Words wa, wb;\$|$WAITHEAD }\n      Words wa, wb;|"
         "s|$DONE_AT|\1\n    if constexpr (MODE == MODE_CROSS \&\& DIGLDS) { __builtin_amdgcn_s_waitcnt(0x0F70); $FLAG }|")
  [ $v != depsc1S ] && edits+=("s|word_at\((src[01]\[r\]), \(uint32_t\)v\)|$SC1LOAD|")
  [ $v != depsc1L ] && edits+=("$SC1STORE")
  tools/ab_build.sh $v --driver tools/probe_overlap.cpp "${edits[@]}"
done

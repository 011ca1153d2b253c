// libbreedhip.so -- the step/reset launches: bh_cross, bh_cross_packed, bh_gather, bh_gather_packed, bh_pack,
// bh_gebv (dispatch over the GEBV routes), bh_reward_max.
#include <cstdlib>
#include "ctx.h"
#include "mutation.h"
#include "packed_kernels.h"
#include "row_kernels.h"

using namespace bhh;

namespace {

// waves per cross in the standalone mask kernel: long rows are split over the workgroup (LDS per
// cross is M/4 bytes, and the Threefry trips of one cross are independent)
int waves_for(const bh_ctx *c) { return c->M <= 16384 ? 1 : 4; }
// waves per row in the row kernel: only a few long rows are worth four waves each
int row_waves_for(const bh_ctx *c, int64_t n_rows) { return (c->M > 16384 && n_rows < 2048) ? 4 : 1; }

size_t mask_group_lds(const bh_ctx *c) { return (size_t)(c->nwh + 1) * 32; }  // S[nwh + 1][4] u64

// rows one wave streams at once (the effect vectors are loaded once per trip for all of them)
constexpr int kRowsPerWaveCross = 2, kRowsPerWavePlain = 4;

template <int WAVES, int TT, int MODE, int R>
int launch_row(const bh_ctx *c, bh::RowParams p) {
  const int per_block = R * 4 / WAVES;
  p.n_blocks = (p.n_rows + per_block - 1) / per_block;
  if (R == 2 && c->M > 16384)
    hipLaunchKernelGGL((bh::row_kernel<WAVES, TT, MODE, R, R == 2>), dim3((unsigned)p.n_blocks), dim3(256), 0,
                       c->stream, p);
  else
    hipLaunchKernelGGL((bh::row_kernel<WAVES, TT, MODE, R, false>), dim3((unsigned)p.n_blocks), dim3(256), 0,
                       c->stream, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(BH_ERR_HIP, "row_kernel launch: %s", hipGetErrorString(e));
  return BH_OK;
}

template <int WAVES, int MODE, int R>
int launch_row_tt(const bh_ctx *c, const bh::RowParams &p, int tt) {
  switch (tt) {
    case 0: return launch_row<WAVES, 0, MODE, R>(c, p);
    case 1: return launch_row<WAVES, 1, MODE, R>(c, p);
    case 2: return launch_row<WAVES, 2, MODE, R>(c, p);
    case 4: return launch_row<WAVES, 4, MODE, R>(c, p);
    default: return launch_row<WAVES, 8, MODE, R>(c, p);
  }
}

template <int MODE>
int launch_row_mode(const bh_ctx *c, const bh::RowParams &p, int tt) {
  if (row_waves_for(c, p.n_rows) == 4) return launch_row_tt<4, MODE, 1>(c, p, tt);
  // few rows: fewer per wave keeps more waves in flight
  if (p.n_rows < 4096) return launch_row_tt<1, MODE, 1>(c, p, tt);
  if (MODE == bh::MODE_CROSS || p.n_rows < 16384) return launch_row_tt<1, MODE, kRowsPerWaveCross>(c, p, tt);
  return launch_row_tt<1, MODE, kRowsPerWavePlain>(c, p, tt);
}

// `waves` = waves per cross in the mask role (1 or 4)
// the key window of a call with B keyed crosses: (offset, total) in crosses; total == B without sharding
struct KeyWindow {
  int64_t off, total;
};

bh::MaskParams mask_params(const bh_ctx *c, int buf, uint32_t key0, uint32_t key1, int64_t B, int waves, bool dh,
                           KeyWindow kw) {
  const int groups = 4 / waves;
  bh::MaskParams m{};
  m.Z = c->d_Z[buf];
  m.thrA = c->d_thrA;
  m.thrB = c->d_thrB;
  m.ctrB = c->d_ctrB;
  m.alwA = c->d_alwA;
  m.alwB = c->d_alwB;
  m.thr_bytes = (uint32_t)(c->nthr * 4);
  m.zpitch = (int64_t)16 * c->zw;
  m.M = (int32_t)c->M;
  m.half = c->half;
  m.nwh = c->nwh;
  m.nw = c->nw;
  m.n_rows = (int32_t)B;
  m.n_blocks = (int32_t)((B + groups - 1) / groups);
  m.group_lds = (int32_t)mask_group_lds(c);
  m.key0 = key0;
  m.key1 = key1;
  m.nkeys = (uint32_t)(dh ? kw.total + 1 : 2 * kw.total + 1);
  m.key_offset = (uint32_t)kw.off;
  m.err = c->d_err;
  m.dh = dh;
  m.layout = c->layout;
  return m;
}

template <typename K>
int allow_big_lds(bh_ctx *c, K kern, size_t lds) {
  if (lds <= 64 * 1024) return BH_OK;
  // once per kernel instance and context: the attribute call is a driver call, and the headline launch (59 KB of digit
  // copy on top of its mask groups' LDS) would pay it on every step of the host path
  const void *f = reinterpret_cast<const void *>(kern);
  if (std::find(c->big_lds_kernels.begin(), c->big_lds_kernels.end(), f) != c->big_lds_kernels.end()) return BH_OK;
  hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
  if (e != hipSuccess) return fail(BH_ERR_HIP, "hipFuncSetAttribute(dynamic LDS): %s", hipGetErrorString(e));
  c->big_lds_kernels.push_back(f);
  return BH_OK;
}

// rows whose site words no longer fit one workgroup's LDS next to everything else (M / 4 bytes: 64 KB here)
constexpr int64_t kLdsMaskMaxMarkers = (int64_t)1 << 18;
int ensure_sites(bh_ctx *c, int64_t B);

// site words -> crossover masks (mask_stream.h): a wave per cross when there are many, four when there are few
int launch_mask_finish(bh_ctx *c, const bh::MaskParams &m) {
  if (m.n_rows <= 0) return BH_OK;
  if (m.n_rows < 4096 && m.nw >= 256)
    hipLaunchKernelGGL(bh::mask_finish4_kernel, dim3((unsigned)m.n_rows), dim3(256), 0, c->stream, m);
  else
    hipLaunchKernelGGL(bh::mask_finish_kernel, dim3((unsigned)((m.n_rows + 3) / 4)), dim3(256), 0, c->stream, m);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(BH_ERR_HIP, "mask_finish_kernel launch: %s", hipGetErrorString(e));
  return BH_OK;
}

// crossover masks of B crosses keyed off `key` into d_Z[buf], in stream order (not overlapped)
int launch_masks(bh_ctx *c, int buf, uint32_t key0, uint32_t key1, int64_t B, bool dh, KeyWindow kw) {
  if (c->M > kLdsMaskMaxMarkers) {
    // the two phases of the mask role as two kernels with the site words in memory (mask_stream.h): any number
    // of workgroups per cross draw the sites, one wave per cross runs the prefix-XOR along the row
    bh::MaskParams m = mask_params(c, buf, key0, key1, B, 4, dh, kw);
    if (int rc = ensure_sites(c, B)) return rc;
    m.sites = c->d_S;
    m.spitch = c->nwh + 1;
    const int64_t tiles = (c->nwh + bh::kSiteTileTrips - 1) / bh::kSiteTileTrips;
    if (B * tiles > 0x7FFFFFFF) return fail(BH_ERR_UNSUPPORTED, "too many crosses x markers for one mask launch");
    hipLaunchKernelGGL(bh::mask_sites_kernel, dim3((unsigned)(B * tiles)), dim3(256), 0, c->stream, m);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(BH_ERR_HIP, "mask_sites_kernel launch: %s", hipGetErrorString(e));
    return launch_mask_finish(c, m);
  }
  const int waves = waves_for(c);
  const bh::MaskParams m = mask_params(c, buf, key0, key1, B, waves, dh, kw);
  const size_t lds = (size_t)m.group_lds * (4 / waves);
  if (waves == 1) {
    hipLaunchKernelGGL(bh::mask_kernel<1>, dim3((unsigned)m.n_blocks), dim3(256), lds, c->stream, m);
  } else {
    if (int rc = allow_big_lds(c, bh::mask_kernel<4>, lds)) return rc;
    hipLaunchKernelGGL(bh::mask_kernel<4>, dim3((unsigned)m.n_blocks), dim3(256), lds, c->stream, m);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(BH_ERR_HIP, "mask_kernel launch: %s", hipGetErrorString(e));
  return BH_OK;
}

// one launch: waves 0-3 of every workgroup stream `rp` (generation g, R rows per wave), waves 4-7
// fill `mp` (generation g+1) with MW waves per cross.  Only for rows one wave handles.
// Grid of the two-role launch.  With somewhat more mask units than row workgroups (shared keys at
// 2k x 100k: 2048 against 1024) a grid as long as the mask work leaves a tail of mask-only workgroups
// behind the streaming (+220 us measured); the workgroups of a grid as long as the row work take the
// mask units in turns instead (cross_kernel).  With many more mask units than row workgroups the
// launch is mask-bound whatever is done, and short workgroups balance better over the CUs than
// long-lived ones (2.49 ms against 2.10 ms at 8192 independent 100k-marker crosses): no turns then.
// Only for four mask waves per cross: in the one-wave-per-cross kernels the loop costs the registers
// that keep eight waves per SIMD resident.
constexpr int kMaskTurnsMax = 4;  // (1 = never take turns)
inline int two_role_grid(int row_blocks, int mask_blocks) {
  if (mask_blocks <= row_blocks) return row_blocks;
  return mask_blocks <= kMaskTurnsMax * row_blocks ? row_blocks : mask_blocks;
}

template <int TT, int MW, int R>
int launch_cross_tt(bh_ctx *c, bh::RowParams rp, const bh::MaskParams &mp) {
  rp.n_blocks = (rp.n_rows + 4 * R - 1) / (4 * R);
  const size_t lds = (size_t)mp.group_lds * (4 / MW);
  const int grid = MW == 4 ? two_role_grid(rp.n_blocks, mp.n_blocks) : std::max(rp.n_blocks, mp.n_blocks);
  if (R == 2 && c->M > 16384)
    hipLaunchKernelGGL((bh::cross_kernel<TT, MW, R, R == 2>), dim3((unsigned)grid), dim3(512), lds, c->stream, rp, mp);
  else
    hipLaunchKernelGGL((bh::cross_kernel<TT, MW, R, false>), dim3((unsigned)grid), dim3(512), lds, c->stream, rp, mp);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(BH_ERR_HIP, "cross_kernel launch: %s", hipGetErrorString(e));
  return BH_OK;
}

template <int MW, int R>
int launch_cross_mw(bh_ctx *c, const bh::RowParams &rp, const bh::MaskParams &mp, int tt) {
  switch (tt) {
    case 0: return launch_cross_tt<0, MW, R>(c, rp, mp);
    case 1: return launch_cross_tt<1, MW, R>(c, rp, mp);
    case 2: return launch_cross_tt<2, MW, R>(c, rp, mp);
    case 4: return launch_cross_tt<4, MW, R>(c, rp, mp);
    default: return launch_cross_tt<8, MW, R>(c, rp, mp);
  }
}

// ---- packed (bit-plane) rows: packed_kernels.h ------------------------------------------------
constexpr int kRowsPerWavePacked = 2;  // rows one wave streams at once when there are many
constexpr int kRowsPerWaveLong = 1;    // rows a wave interleaves in the long-row stream

int tt_for(int tcount) { return tcount <= 0 ? 0 : tcount == 1 ? 1 : tcount == 2 ? 2 : tcount <= 4 ? 4 : 8; }

// bh_gather_packed's arguments -> the launch it becomes (shared by the call itself and by the gather asked for ahead)
struct GatherPlan {
  bh::PackedParams p;
  bool fused = false, mfma_after = false;
  int turns = 0;
};
GatherPlan plan_gather(const bh_ctx *c, const uint16_t *d_src, int64_t src_rows, const int32_t *d_idx, int64_t n_rows,
                       uint16_t *d_out_packed, uint8_t *d_out_bytes, float *d_gebv);
int run_gather(bh_ctx *c, const GatherPlan &g, const uint16_t *d_src, int64_t src_rows, const int32_t *d_idx,
               int64_t n_rows, float *d_gebv);

template <int TT, int MW, int R>
int launch_pcross_tt(bh_ctx *c, bh::PackedParams rp, const bh::MaskParams &mp) {
  const int per_wave = R * std::max(1, rp.seq);
  rp.n_blocks = (rp.n_rows + 4 * per_wave - 1) / (4 * per_wave);
  // stream-per-lane mask role: no LDS, its waves share the mask work whatever the grid is
  size_t lds = mp.sites ? 0 : (size_t)mp.group_lds * (4 / MW);
  int grid = mp.sites ? rp.n_blocks
                      : (MW == 4 ? two_role_grid(rp.n_blocks, mp.n_blocks) : std::max(rp.n_blocks, mp.n_blocks));
  bh::PackedParams gp{};  // n_blocks = 0: nothing rides
  if (MW == 4 && TT >= 1 && TT <= 2 && !mp.sites && c->gather_ahead.pending) {
    // a gather asked for ahead (the founders of the reset behind a horizon step) rides on this launch if it would
    // have become the same kind of stream: packed rows out, its GEBV fused with this many traits, rows in twos or not
    const auto &ga = c->gather_ahead;
    const GatherPlan g = plan_gather(c, ga.src, ga.src_rows, ga.idx, ga.n_rows, ga.outp, ga.outb, ga.gebv);
    if (g.fused && g.turns <= 1 && ga.outp && tt_for(g.p.tcount) == TT && (ga.n_rows >= 4096 ? 2 : 1) == R &&
        !(rp.outb && rp.obs_nt) && !(g.p.outb && g.p.obs_nt)) {
      gp = g.p;
      gp.n_blocks = (gp.n_rows + 8 * R - 1) / (8 * R);  // all eight waves of a rider stream rows
      grid += gp.n_blocks;
      c->gather_ahead.pending = false;
    }
  }
  bh::FoundersParams fp{};
  if (MW == 4 && c->founders_pending && c->fp_pending.P <= 1024 && (size_t)c->fp_pending.P * 16 <= 48 * 1024) {
    fp = c->fp_pending;  // the draw rides on this launch: E more workgroups
    c->founders_pending = false;
    grid += fp.E;
    lds = std::max(lds, (size_t)fp.P * 16);
  }
  if (MW == 4 && TT == 1 && c->T == 1 && !mp.sites) {
    // one trait, rows short enough: the row waves keep a copy of its digit planes in LDS (packed_stream_role, DIGLDS)
    // -- where two workgroups per CU still fit beside everything else (48 bytes per eight markers: 59 KB at 10k)
    const size_t dig = ((size_t)48 * (c->nvec + 1) + 1023) / 1024 * 1024;
    const size_t at = (lds + 15) / 16 * 16;
    if (at + dig <= 76 * 1024) {
      rp.dig_lds = (int32_t)at + 1;
      if (gp.n_blocks > 0) gp.dig_lds = rp.dig_lds;
      lds = at + dig;
    }
  }
  // an instance per Threefry layout (and one for riders): a kernel that carried both layouts' mask loops, the riders'
  // paths and both store policies was 13.5 k instructions, beyond the reach of a branch (tests/test_isa.py)
  auto go = [&](auto kernel) -> int {
    if (int rc = allow_big_lds(c, kernel, lds)) return rc;
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(512), lds, c->stream, rp, mp, fp, gp);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(BH_ERR_HIP, "packed_cross_kernel launch: %s", hipGetErrorString(e));
    return BH_OK;
  };
  const bool part = mp.layout == bh::LAYOUT_PARTITIONABLE;
  int rc_launch;
  if constexpr (MW == 4 && TT >= 1 && TT <= 2) {
    if (gp.n_blocks > 0)
      rc_launch = part ? go(bh::packed_cross_kernel<TT, MW, R, true, bh::LAYOUT_PARTITIONABLE>)
                       : go(bh::packed_cross_kernel<TT, MW, R, true, bh::LAYOUT_ORIGINAL>);
    else
      rc_launch = part ? go(bh::packed_cross_kernel<TT, MW, R, false, bh::LAYOUT_PARTITIONABLE>)
                       : go(bh::packed_cross_kernel<TT, MW, R, false, bh::LAYOUT_ORIGINAL>);
  } else if (mp.sites) {
    rc_launch = go(bh::packed_cross_kernel<TT, MW, R, false, -1>);
  } else {
    rc_launch = part ? go(bh::packed_cross_kernel<TT, MW, R, false, bh::LAYOUT_PARTITIONABLE>)
                     : go(bh::packed_cross_kernel<TT, MW, R, false, bh::LAYOUT_ORIGINAL>);
  }
  if (rc_launch) return rc_launch;
  if (mp.sites) return launch_mask_finish(c, mp);  // site words -> crossover masks
  return BH_OK;
}

// the launch for many long rows: stream role in turns + stream-per-lane mask role, then the site words -> masks
constexpr int kLongMaskWaves = 8;  // mask waves per workgroup of four row waves
int launch_plong(bh_ctx *c, bh::PackedParams rp, const bh::MaskParams &mp) {
  const int per_wave = kRowsPerWaveLong * std::max(1, rp.seq);
  rp.n_blocks = (rp.n_rows + 4 * per_wave - 1) / (4 * per_wave);
  hipLaunchKernelGGL((bh::packed_long_kernel<kRowsPerWaveLong, kLongMaskWaves>), dim3((unsigned)rp.n_blocks),
                     dim3(64 * (4 + kLongMaskWaves)), 0, c->stream, rp, mp);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(BH_ERR_HIP, "packed_long_kernel launch: %s", hipGetErrorString(e));
  // site words -> masks, in stream order (a stream of its own beside the GEBV pass that follows was measured: 811
  // against 818 us per step at 8192 x 100k -- both want the same memory system)
  return launch_mask_finish(c, mp);
}

template <int MW, int R>
int launch_pcross_mw(bh_ctx *c, const bh::PackedParams &rp, const bh::MaskParams &mp, int tt) {
  switch (tt) {
    case 0: return launch_pcross_tt<0, MW, R>(c, rp, mp);
    case 1: return launch_pcross_tt<1, MW, R>(c, rp, mp);
    case 2: return launch_pcross_tt<2, MW, R>(c, rp, mp);
    case 4: return launch_pcross_tt<4, MW, R>(c, rp, mp);
    default: return launch_pcross_tt<8, MW, R>(c, rp, mp);
  }
}

template <int TT, int MODE, int R>
int launch_prow(const bh_ctx *c, bh::PackedParams p) {
  const int per_wave = R * std::max(1, p.seq);
  p.n_blocks = (p.n_rows + 4 * per_wave - 1) / (4 * per_wave);
  hipLaunchKernelGGL((bh::packed_row_kernel<TT, MODE, R>), dim3((unsigned)p.n_blocks), dim3(256), 0, c->stream, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(BH_ERR_HIP, "packed_row_kernel launch: %s", hipGetErrorString(e));
  return BH_OK;
}

template <int MODE, int R>
int launch_prow_r(const bh_ctx *c, const bh::PackedParams &p, int tt) {
  switch (tt) {
    case 0: return launch_prow<0, MODE, R>(c, p);
    case 1: return launch_prow<1, MODE, R>(c, p);
    case 2: return launch_prow<2, MODE, R>(c, p);
    case 4: return launch_prow<4, MODE, R>(c, p);
    default: return launch_prow<8, MODE, R>(c, p);
  }
}

template <int MODE>
int launch_prow_mode(const bh_ctx *c, const bh::PackedParams &p, int tt) {
  // few rows: one per wave keeps more waves in flight
  return p.n_rows < 4096 ? launch_prow_r<MODE, 1>(c, p, tt) : launch_prow_r<MODE, kRowsPerWavePacked>(c, p, tt);
}

// few crosses need masks (they are shared by all envs, or the batch is small): put four waves on
// each so that the mask role is as short as the streaming beside it.  "Few" is against the row work beside them: with
// shared keys and eight envs or more there is at most one keyed cross per row workgroup (eight rows) however large the
// batch -- 64 envs x 1024 took the one-wave form (and with it the plain f64 row loop) at 672 us; four-wave groups and
// the pipelined stream: 2.0x less
int mask_waves_for_runahead(const bh_ctx *c, int64_t B, int64_t rows) {
  return (B * 4 <= 2048 || c->M > 16384 || B * 8 <= rows) ? 4 : 1;
}

// How the pipelined stream stores the byte observation (PackedParams::obs_nt), measured at 10k markers
// (tools/time_stream_obs.py, profiles/r04_ab_obs_stores.txt): while two alternating observation blocks and the state
// fit the 256 MB Infinity Cache, write-through stores (sc1: the line leaves the XCD's L2 at once, nothing is left to
// write back when the kernel ends) take 0.6 us off the launch and 0.4 off the gap behind it; beyond, write-through is
// the slowest form (8192 rows: 67 us against 43) and non-temporal the fastest by far (16384 rows: 80 us against 140 for
// plain stores, 32768: 163 against 280).
int obs_store_nt(const bh_ctx *c, int64_t rows) { return rows * c->pitch > (int64_t)96 << 20; }


// make d_Z[buf] hold at least `bytes`; growing it waits for everything that may still touch it
int ensure_z(bh_ctx *c, int buf, int64_t bytes) {
  if (c->z_cap[buf] >= bytes) return BH_OK;
  BH_HIP(hipStreamSynchronize(c->stream));
  if (c->d_Z[buf]) BH_HIP(hipFree(c->d_Z[buf]));
  c->d_Z[buf] = nullptr;
  c->z_cap[buf] = 0;
  if (c->pf.valid && c->pf.buf == buf) c->pf.valid = false;
  BH_HIP(hipMalloc(&c->d_Z[buf], (size_t)bytes));
  c->z_cap[buf] = bytes;
  return BH_OK;
}

// Long bit-plane rows without GEBV in the stream (it follows as the matrix-core pass): a wave streams
// its rows one after the other, so that about kLongStreams rows are being written at any time
// (packed_stream_role).  0 = the ordinary stream.
constexpr int kLongStreams = 4096;
int long_row_turns(const bh_ctx *c, int64_t rows) {
  if (c->M <= 16384 || rows < 2 * kLongStreams) return 0;
  return (int)std::min<int64_t>(16, rows / kLongStreams);
}

// site words of B crosses for mask_stream_role: (B, nwh + 1, 4) u64, zero where never written
int ensure_sites(bh_ctx *c, int64_t B) {
  const int64_t bytes = B * (c->nwh + 1) * 32;
  if (c->s_cap >= bytes) return BH_OK;
  BH_HIP(hipStreamSynchronize(c->stream));
  if (c->d_S) BH_HIP(hipFree(c->d_S));
  c->d_S = nullptr;
  c->s_cap = 0;
  BH_HIP(hipMalloc(&c->d_S, (size_t)bytes));
  BH_HIP(hipMemsetAsync(c->d_S, 0, (size_t)bytes, c->stream));
  c->s_cap = bytes;
  return BH_OK;
}

// mutation (mutation.h): draw the flips of the call's keyed gametes and XOR them into the offspring rows the cross
// launch just wrote.  B keyed crosses; `rows` offspring rows of `n_out` per env.
int mutate_rows(bh_ctx *c, uint32_t key0, uint32_t key1, int64_t nkeys, KeyWindow kw, int64_t B, bool dh, bool shared,
                int64_t rows, int64_t n_out, uint16_t *d_out_packed, uint8_t *d_out_bytes) {
  const int32_t half = c->layout == bh::LAYOUT_PARTITIONABLE ? (int32_t)c->M : (int32_t)((c->M + 1) / 2);
  const int32_t wl = (half + 63) / 64;
  const int64_t fpitch = 2 * (int64_t)wl + 1, streams = dh ? B : 2 * B;
  const int64_t bytes = streams * fpitch * 8;
  if (c->f_cap < bytes) {
    BH_HIP(hipStreamSynchronize(c->stream));
    if (c->d_F) BH_HIP(hipFree(c->d_F));
    c->d_F = nullptr;
    c->f_cap = 0;
    BH_HIP(hipMalloc(&c->d_F, (size_t)bytes));
    c->f_cap = bytes;
  }
  BH_HIP(hipMemsetAsync(c->d_F, 0, (size_t)bytes, c->stream));  // the pad word of every stream
  bh::MutationParams sp{};
  sp.F = c->d_F;
  sp.fpitch = fpitch;
  sp.M = (int32_t)c->M;
  sp.half = half;
  sp.wl = wl;
  sp.n_streams = (int32_t)streams;
  sp.key0 = key0;
  sp.key1 = key1;
  sp.nkeys = (uint32_t)nkeys;
  sp.key_offset = (uint32_t)kw.off;
  sp.dh = dh;
  sp.layout = c->layout;
  sp.thr = c->mut_thr;
  for (int64_t s0 = 0; s0 < streams; s0 += 32768) {  // grid.y holds 65535
    const int64_t ns = std::min<int64_t>(32768, streams - s0);
    bh::MutationParams q = sp;
    q.F = sp.F + s0 * fpitch;
    // stream s0 + y: advance the key index by s0 streams (s0 is even: gamete parity is kept)
    q.key_offset = (uint32_t)(kw.off + (dh ? s0 : s0 / 2));
    // (a wave takes two chunks per turn; few workgroups per stream: each starts with the three Threefry calls of its key)
    hipLaunchKernelGGL(bh::mutation_sites_kernel, dim3((unsigned)std::min(8, (wl + 7) / 8), (unsigned)ns), dim3(256), 0,
                       c->stream, q);
    BH_HIP(hipGetLastError());
  }
  bh::MutationApplyParams ap{};
  ap.F = c->d_F;
  ap.fpitch = fpitch;
  ap.M = (int32_t)c->M;
  ap.half = half;
  ap.wl = wl;
  ap.n_rows = (int32_t)rows;
  ap.n_out = (int32_t)n_out;
  ap.shared = shared;
  ap.dh = dh;
  ap.outp = d_out_packed;
  ap.ppitch = 8 * (int64_t)c->pw;
  ap.outb = d_out_bytes;
  ap.bpitch = c->pitch;
  const int64_t threads = streams * 2 * wl;  // one per flip word
  if (threads > 0 && rows > 0) {
    if ((threads + 255) / 256 > 0x7FFFFFFF) return fail(BH_ERR_UNSUPPORTED, "mutation: too many crosses for one launch");
    hipLaunchKernelGGL(bh::mutation_apply_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, c->stream, ap,
                       (int32_t)streams);
    BH_HIP(hipGetLastError());
  }
  return BH_OK;
}


bh::RowParams base_params(const bh_ctx *c) {
  bh::RowParams p{};
  p.eff = c->d_eff;
  p.dig = c->d_dig16;
  p.scale = c->d_scale;
  p.pitch = c->pitch;
  p.effPitch = c->effPitch;
  p.zpitch = (int64_t)16 * c->zw;
  p.nvec = c->nvec;
  p.T = (int32_t)c->T;
  return p;
}

bh::PackedParams packed_params(const bh_ctx *c) {
  bh::PackedParams p{};
  p.eff = c->d_eff;
  p.dig = c->d_dig16;
  p.scale = c->d_scale;
  p.pitch = c->pitch;
  p.effPitch = c->effPitch;
  p.zpitch = (int64_t)16 * c->zw;
  p.nvec = c->nvec;
  p.nvp = 8 * c->pw;
  p.T = (int32_t)c->T;
  p.spare = c->d_spare;
  p.err = c->d_err;
  return p;
}

// from this many traits on, the int8-MFMA GEBV pass beats GEBV fused into the row stream
constexpr int kMfmaGebvMinTraits = 3;
bool use_mfma_gebv(const bh_ctx *c, int64_t rows) { return c->T >= kMfmaGebvMinTraits && rows >= 32; }
// bit-plane rows: the matrix-core pass (gebv_bits.h) also for one or two traits once the rows are long --
// at 100k markers it takes a quarter of the f64 stream's time (profiles/r02_gebv_bits.txt)
constexpr int kBitsGebvMinMarkers = 16385;
bool use_bits_gebv(const bh_ctx *c, int64_t rows) {
  return rows >= 32 && (c->T >= kMfmaGebvMinTraits || c->M >= kBitsGebvMinMarkers);
}

// GEBV of rows already in memory, for trait chunks [t_begin, T)
int launch_pack(bh_ctx *c, const bh::PackParams &p) {
  for (int64_t r0 = 0; r0 < p.n_rows; r0 += 65535) {  // grid.y holds 65535
    bh::PackParams q = p;
    q.src = p.src + r0 * p.pitch;
    q.dst = p.dst + r0 * p.nvp;
    q.n_rows = (int32_t)std::min<int64_t>(65535, p.n_rows - r0);
    hipLaunchKernelGGL(bh::pack_kernel, dim3((unsigned)((p.nvp + 2047) / 2048), (unsigned)q.n_rows), dim3(256), 0, c->stream, q);
    BH_HIP(hipGetLastError());
  }
  return BH_OK;
}

constexpr int64_t kGebvViaPlanesMin = 1ll << 25;  // rows x markers from which byte-layout rows take the one-pass plane route
int gebv_chunks(bh_ctx *c, const uint8_t *d_pop, int64_t rows, float *d_gebv, int t_begin) {
  // large byte-layout populations: one pass that turns the rows into bit planes inside LDS and runs the matrix-core
  // digit GEMMs of gebv_bits.h on those (gebv_bytes.h) -- 2048 x 100k x 8: 158 us (gebv_mfma_kernel) -> 115 us (round 2:
  // planes through HBM) -> this; long rows of one or two traits: instead of the f64 stream
  if (t_begin == 0 && use_bits_gebv(c, rows) && rows * c->M >= kGebvViaPlanesMin) return gebv_bytes(c, d_pop, rows, d_gebv);
  if (t_begin == 0 && use_mfma_gebv(c, rows)) return gebv_mfma(c, d_pop, rows, d_gebv);
  for (int t0 = t_begin; t0 < c->T; t0 += kMaxFusedTraits) {
    bh::RowParams p = base_params(c);
    p.src = d_pop;
    p.gebv = d_gebv;
    p.n_rows = (int32_t)rows;
    p.t0 = t0;
    p.tcount = (int32_t)std::min<int64_t>(kMaxFusedTraits, c->T - t0);
    p.eff = c->d_eff + (int64_t)t0 * c->effPitch;
    int rc = launch_row_mode<bh::MODE_GEBV>(c, p, tt_for(p.tcount));
    if (rc) return rc;
  }
  return BH_OK;
}

// GEBV of packed rows, trait chunks [t_begin, T), fused f64 path (packed_kernels.h)
int packed_gebv_chunks(bh_ctx *c, const uint16_t *d_rows, int64_t rows, float *d_gebv, int t_begin) {
  for (int t0 = t_begin; t0 < c->T; t0 += kMaxFusedTraits) {
    bh::PackedParams p = packed_params(c);
    p.src = d_rows;
    p.gebv = d_gebv;
    p.n_pool = (int32_t)rows;
    p.n_rows = (int32_t)rows;
    p.t0 = t0;
    p.tcount = (int32_t)std::min<int64_t>(kMaxFusedTraits, c->T - t0);
    p.eff = c->d_eff + (int64_t)t0 * c->effPitch;
    if (int rc = launch_prow_mode<bh::MODE_GATHER>(c, p, tt_for(p.tcount))) return rc;
  }
  return BH_OK;
}

GatherPlan plan_gather(const bh_ctx *c, const uint16_t *d_src, int64_t src_rows, const int32_t *d_idx, int64_t n_rows,
                       uint16_t *d_out_packed, uint8_t *d_out_bytes, float *d_gebv) {
  GatherPlan g;
  bh::PackedParams &p = g.p;
  p = packed_params(c);
  p.src = d_src;
  p.outp = d_out_packed;
  p.outb = d_out_bytes;
  p.obs_nt = obs_store_nt(c, n_rows);
  p.idx = d_idx;
  p.gebv = d_gebv;
  p.n_pool = (int32_t)src_rows;
  p.n_rows = (int32_t)n_rows;
  const bool writes = d_out_packed || d_out_bytes;
  g.mfma_after = d_gebv && use_bits_gebv(c, n_rows) &&
                 (c->T >= kMfmaGebvMinTraits || !writes || (d_out_packed && long_row_turns(c, n_rows) > 1));
  g.fused = d_gebv && c->T <= kMaxFusedTraits && !g.mfma_after;
  p.tcount = g.fused ? (int32_t)c->T : 0;
  g.turns = (p.tcount == 0 && d_out_packed) ? long_row_turns(c, n_rows) : 0;
  // (long rows without GEBV in the stream: the pipelined stream, not the plain loop -- one row per wave unless in turns)
  p.seq = g.turns > 1 ? g.turns : (p.tcount == 0 && d_out_packed && d_out_bytes && c->M > 16384 ? 1 : 0);
  return g;
}

int run_gather(bh_ctx *c, const GatherPlan &g, const uint16_t *d_src, int64_t src_rows, const int32_t *d_idx,
               int64_t n_rows, float *d_gebv) {
  const bh::PackedParams &p = g.p;
  if (p.outp && !p.outb && !d_gebv && n_rows <= 65535 && g.turns <= 1) {
    // bit planes out and nothing else (the winners' rows of a population-sharded generation): plain copies, a row cut
    // over as many waves as it has 4 KB pieces (packed_copy_rows_kernel)
    bh::CopyRowsParams q{};
    q.src = reinterpret_cast<const uint8_t *>(d_src);
    q.dst = reinterpret_cast<uint8_t *>(p.outp);
    q.idx = d_idx;
    q.row_bytes = (int64_t)16 * c->pw;
    q.n_pool = (int32_t)src_rows;
    q.n_rows = (int32_t)n_rows;
    hipLaunchKernelGGL(bh::packed_copy_rows_kernel, dim3((unsigned)((q.row_bytes / 16 + 1023) / 1024), (unsigned)n_rows), dim3(256), 0,
                       c->stream, q);
    BH_HIP(hipGetLastError());
    return BH_OK;
  }
  if (!p.outp && p.outb && p.tcount == 0) {
    // byte rows out and nothing else in the stream (unpack; a lazy env's observation): rows cut over waves
    bh::ExpandParams q{};
    q.src = d_src;
    q.idx = d_idx;
    q.dst = p.outb;
    q.pitch = c->pitch;
    q.nvec = c->nvec;
    q.nvp = 8 * c->pw;
    q.n_pool = (int32_t)src_rows;
    q.n_rows = (int32_t)n_rows;
    q.nt = p.obs_nt;
    for (int64_t r0 = 0; r0 < n_rows; r0 += 65535) {  // grid.y holds 65535
      q.row0 = (int32_t)r0;
      hipLaunchKernelGGL(bh::packed_expand_kernel, dim3((unsigned)((q.nvec + 2047) / 2048), (unsigned)std::min<int64_t>(65535, n_rows - r0)),
                         dim3(256), 0, c->stream, q);
      BH_HIP(hipGetLastError());
    }
  } else if (p.tcount > 0 || p.outp || p.outb) {
    if (int rc = g.turns > 1 ? launch_prow_r<bh::MODE_GATHER, kRowsPerWaveLong>(c, p, 0)
                             : launch_prow_mode<bh::MODE_GATHER>(c, p, tt_for(p.tcount)))
      return rc;
  }
  if (d_gebv && !g.fused) {
    // the matrix-core pass straight over the source rows, through the index (gebv_bits.h)
    if (g.mfma_after) return gebv_bits(c, d_src, src_rows, d_idx, n_rows, d_gebv);
    // few rows or more than eight traits: f64 passes over the packed rows, through the index
    for (int t0 = 0; t0 < c->T; t0 += kMaxFusedTraits) {
      bh::PackedParams q = p;
      q.outp = nullptr;
      q.outb = nullptr;
      q.t0 = t0;
      q.tcount = (int32_t)std::min<int64_t>(kMaxFusedTraits, c->T - t0);
      q.eff = c->d_eff + (int64_t)t0 * c->effPitch;
      if (int rc = launch_prow_mode<bh::MODE_GATHER>(c, q, tt_for(q.tcount))) return rc;
    }
  }
  return BH_OK;
}

}  // namespace

namespace bhh {

// a gather asked for ahead that found no launch to ride on is made now, in stream order
int flush_gather(bh_ctx *c) {
  if (!c->gather_ahead.pending) return BH_OK;
  c->gather_ahead.pending = false;
  const auto ga = c->gather_ahead;
  const GatherPlan g = plan_gather(c, ga.src, ga.src_rows, ga.idx, ga.n_rows, ga.outp, ga.outb, ga.gebv);
  return run_gather(c, g, ga.src, ga.src_rows, ga.idx, ga.n_rows, ga.gebv);
}

}  // namespace bhh

extern "C" {

// bh_cross and bh_cross_packed: `packed` selects the layout of the pool and of d_out_packed;
// d_out_bytes is the byte-layout offspring (required unless packed, optional when packed).
//
// Which kernels a call becomes.  M markers, rows = n_envs * n_out offspring, B = crosses that own keys (n_out with
// shared keys, rows otherwise), Bn = B of the NEXT call (= B unless bh_hint_next_cross said otherwise), T traits.
//
//   this generation's masks   made ahead by the previous call's launch (key, B, double-haploid flag and key window all
//                             as predicted)  -> nothing to do;  otherwise launch_masks in stream order: mask_kernel<1>
//                             (M <= 16384), mask_kernel<4> (M <= 2^18), mask_sites_kernel + mask_finish_kernel (beyond)
//   long_stream               packed && rows >= 8192 && M > 16384 && the mask role would take four waves per cross
//                             (Bn <= 512 or M > 16384) or there is no run-ahead
//   GEBV (when asked for)     fused in the row role: T <= 8, no mutation, and not `mfma_after`;
//                             mfma_after (a matrix-core pass behind the launch): bit planes: T >= 3 and rows >= 32, or
//                             T <= 2 on the long_stream; byte layout: T >= 3 and rows >= 32;
//                             otherwise (T > 8 without MFMA, mutation): f64 passes over the offspring
//   run_ahead (mask role in   prefetch on, Bn > 0, bit planes or one-wave byte rows (not: a few very long byte rows,
//   the same launch)          four waves each), and M <= 2^18 unless long_stream (beyond, only that launch's mask role
//                             keeps no site words in LDS)
//   the launch, bit planes    run_ahead: mw = 1 (Bn > 512 and M <= 16384): packed_cross_kernel<TT, 1, R>;  mw = 4 and
//                             long_stream without fused GEBV: packed_long_kernel<1, 8> (LDS-DMA row stream, stream-per-
//                             lane mask role) + mask_finish_kernel;  else packed_cross_kernel<TT, 4, R> (a deferred
//                             founder draw of <= 1024 rows rides on it);  R = 2 rows per wave from 4096 rows on.
//                             no run_ahead: packed_row_kernel<TT, MODE_CROSS, R> (long rows in turns: its DMA stream)
//                             rows of M > 16384 without fused GEBV whose byte rows are written take the pipelined
//                             stream (seq = 1) where they are too few for turns, not the plain loop
//   the launch, byte layout   run_ahead: cross_kernel<TT, mw, R, LONG>;  else row_kernel<W, TT, MODE_CROSS, R, LONG>
//                             (W = 4 waves per row for fewer than 2048 rows of more than 16384 markers)
//   behind the launch         mutation: mutation_sites_kernel + mutation_apply_kernel;  GEBV passes as above;  reward:
//                             noted inside the launch when the GEBV is fused, else reward_max_kernel
static int cross_impl(bh_ctx *c, const char *who, bool packed, const void *d_pool, int64_t n_envs,
                      int64_t pool_rows, const int32_t *d_idx, int64_t n_out, uint32_t sim_key[2], int rng_mode,
                      uint8_t *d_out_bytes, uint16_t *d_out_packed, float *d_gebv, float *d_reward,
                      float *d_reward_next, int64_t reward_pitch) {
  if (int rc = check_ctx(c)) return rc;
  if (!sim_key) return fail(BH_ERR_BAD_ARG, "%s: null sim_key", who);
  if (n_envs < 0 || n_out < 0 || pool_rows < 0) return fail(BH_ERR_BAD_ARG, "%s: negative size", who);
  if ((d_reward || d_reward_next) && reward_pitch < 1) return fail(BH_ERR_BAD_ARG, "%s: reward_pitch must be >= 1", who);
  if (rng_mode < 0 || rng_mode > (BH_RNG_INDEPENDENT | BH_RNG_DOUBLE_HAPLOID))
    return fail(BH_ERR_BAD_ARG, "%s: unknown rng_mode %d", who, rng_mode);
  const bool dh = (rng_mode & BH_RNG_DOUBLE_HAPLOID) != 0;
  const bool shared = (rng_mode & BH_RNG_INDEPENDENT) == 0;
  const int64_t rows = n_envs * n_out;
  const int64_t B = shared ? n_out : rows;
  // independent keys of a shard: the keys the whole batch would split (bh_set_cross_window, or the env window
  // in crosses); shared keys are the same on every shard anyway
  KeyWindow kw{0, B};
  if (!shared) {
    if (c->crosses_total > 0)
      kw = {c->cross_offset, c->crosses_total};
    else if (c->envs_total > 0)
      kw = {c->env_offset * n_out, c->envs_total * n_out};
    if (kw.off + B > kw.total) return fail(BH_ERR_BAD_ARG, "%s: %lld crosses at offset %lld exceed the window of %lld",
                                           who, (long long)B, (long long)kw.off, (long long)kw.total);
  }
  const int64_t nkeys = dh ? kw.total + 1 : 2 * kw.total + 1;
  if (rows > 0x3FFFFFFF || n_envs * pool_rows > 0x7FFFFFFF || 2 * nkeys > 0xFFFFFFFFll)
    return fail(BH_ERR_UNSUPPORTED, "%s: too many rows for 32-bit indexing", who);
  if (rows > 0) {
    if (!d_pool || !d_idx || (packed ? !d_out_packed : !d_out_bytes))
      return fail(BH_ERR_BAD_ARG, "%s: null device pointer", who);
    if (pool_rows == 0) return fail(BH_ERR_BAD_ARG, "%s: empty parent pool", who);
    if (d_gebv && c->T == 0) return fail(BH_ERR_BAD_ARG, "%s: GEBV requested but the context has no traits", who);
    if (d_reward && !d_gebv) return fail(BH_ERR_BAD_ARG, "%s: the reward needs the GEBV output", who);
    const int64_t row_bytes = packed ? (int64_t)16 * c->pw : c->pitch;
    const uint8_t *pool_lo = static_cast<const uint8_t *>(d_pool), *pool_end = pool_lo + n_envs * pool_rows * row_bytes;
    const uint8_t *out_lo = packed ? reinterpret_cast<const uint8_t *>(d_out_packed) : d_out_bytes;
    const uint8_t *out_end = out_lo + rows * row_bytes;
    if (out_lo < pool_end && pool_lo < out_end) return fail(BH_ERR_BAD_ARG, "%s: the output aliases the pool", who);

    // ---- this generation's masks: made by the previous call's launch, or made now
    const int64_t zbytes = B * 16 * c->zw + 128;  // (+ a line: a row role asks for pw words of every mask row)
    int buf;
    if (c->pf.valid && c->pf.key0 == sim_key[0] && c->pf.key1 == sim_key[1] && c->pf.B == B && c->pf.dh == dh &&
        c->pf.off == kw.off && c->pf.total == kw.total) {
      buf = c->pf.buf;
    } else {  // not predicted: first call, new seed, new batch size
      buf = 1 - c->last_buf;
      if (int rc = ensure_z(c, buf, zbytes)) return rc;
      if (int rc = launch_masks(c, buf, sim_key[0], sim_key[1], B, dh, kw)) return rc;
    }
    c->pf.valid = false;
    c->last_buf = buf;
  }

  // Simulator.cross: keys = split(key, 2B+1); key <- keys[0]
  const bh::u32x2 nk = bh::split_at(sim_key[0], sim_key[1], (uint32_t)nkeys, 0u, c->layout);

  if (rows > 0) {
    const int buf = c->last_buf;
    // GEBV inside the streaming launch (f64 VALU) for few traits; from 3 traits on the int8-MFMA
    // pass over the byte-layout offspring is faster -- when there are byte-layout offspring
    // (bit-plane rows: the pass over the packed offspring, gebv_bits.h)
    // one or two traits leave the stream only where the long-row stream takes over (many long bit-plane rows
    // next to four-wave mask groups): elsewhere the digit products inside the stream are cheaper than a pass
    // the masks made ahead are for the batch the NEXT call will have: the same as this one unless the caller said otherwise
    // (bh_hint_next_cross: a breeding scheme that alternates 200 crosses with 20,000 doubled haploids)
    int64_t Bn = B;
    bool dhn = dh;
    KeyWindow kwn = kw;
    if (c->hint.valid) {
      c->hint.valid = false;
      if (c->crosses_total == 0 && c->envs_total == 0) {  // (with a sharding window set the hint is ignored)
        Bn = (c->hint.rng_mode & BH_RNG_INDEPENDENT) ? c->hint.n_envs * c->hint.n_out : c->hint.n_out;
        dhn = (c->hint.rng_mode & BH_RNG_DOUBLE_HAPLOID) != 0;
        kwn = {0, Bn};
      }
    }
    const bool long_stream = packed && (!c->prefetch || mask_waves_for_runahead(c, Bn, rows) == 4) && long_row_turns(c, rows) > 1;
    const bool mfma_after =
        d_gebv && (packed ? (c->T >= kMfmaGebvMinTraits ? use_bits_gebv(c, rows) : long_stream) : use_mfma_gebv(c, rows));
    // mutation flips alleles after the crossover: the GEBV (and the reward) must see the mutated rows, so they
    // come from passes behind mutate_rows
    const bool mutate = c->mut_thr != 0;
    const bool fused = d_gebv && c->T <= kMaxFusedTraits && !mfma_after && !mutate;
    const int tcount = fused ? (int32_t)c->T : 0;
    // run ahead: the next generation's masks need only the advanced key and the batch size, so
    // the same launch makes them (mask role) while it streams this generation (row role)
    // (beyond kLdsMaskMaxMarkers only the long-row launch can carry the mask role: it keeps no site words in LDS)
    const bool run_ahead = c->prefetch && Bn > 0 && (packed || row_waves_for(c, rows) == 1) &&
                           (c->M <= kLdsMaskMaxMarkers || long_stream);
    bh::MaskParams mp{};
    int mw = 1;
    if (run_ahead) {
      if (int rc = ensure_z(c, 1 - buf, Bn * 16 * c->zw + 128)) return rc;
      mw = mask_waves_for_runahead(c, Bn, rows);
      mp = mask_params(c, 1 - buf, nk.x, nk.y, Bn, mw, dhn, kwn);
    }
    const bool wide = rows >= 4096;  // else one row per wave: more waves in flight
    // the per-env maximum comes out of the same launch when the GEBV does: every row wave notes its value
    // with an atomic max into a buffer that holds -inf -- left so by the launch before (d_reward_next of that
    // call), or set so now
    const bool reward_in_launch = d_reward && fused;
    if (reward_in_launch && !(c->primed_reward == d_reward && c->primed_envs == n_envs && c->primed_pitch == reward_pitch)) {
      hipLaunchKernelGGL(bh::prime_reward_kernel, dim3(1), dim3(256), 0, c->stream, d_reward, (int)n_envs, reward_pitch);
      BH_HIP(hipGetLastError());
    }
    int rc;
    if (packed) {
      bh::PackedParams p = packed_params(c);
      p.src = static_cast<const uint16_t *>(d_pool);
      p.outp = d_out_packed;
      p.outb = d_out_bytes;
      p.obs_nt = obs_store_nt(c, rows);
      p.idx = d_idx;
      p.gebv = d_gebv;
      p.Z = c->d_Z[buf];
      p.n_pool = (int32_t)pool_rows;
      p.n_out = (int32_t)n_out;
      p.n_rows = (int32_t)rows;
      p.rng_shared = shared;
      p.tcount = tcount;
      p.reward = reward_in_launch ? d_reward : nullptr;
      p.reward_next = d_reward_next;
      p.reward_pitch = reward_pitch;
      p.n_envs = (int32_t)n_envs;
      const int turns = (tcount == 0 && (!run_ahead || mw == 4)) ? long_row_turns(c, rows) : 0;
      // (long rows without GEBV in the stream -- three traits and more: the matrix-core pass follows -- take the pipelined
      // stream, not the plain loop, also where they are too few for turns -- where the byte rows are written: 4 x 512 x 100k 243 ->
      // 208 us with them, 143 -> 151 without)
      p.seq = turns > 1 ? turns : (tcount == 0 && d_out_bytes && c->M > 16384 && (!run_ahead || mw == 4) ? 1 : 0);
      if (run_ahead) {
        if (mw == 1)
          rc = wide ? launch_pcross_mw<1, kRowsPerWavePacked>(c, p, mp, tt_for(tcount))
                    : launch_pcross_mw<1, 1>(c, p, mp, tt_for(tcount));
        else if (turns > 1) {
          if (int rc2 = ensure_sites(c, Bn)) return rc2;  // the site words of the batch the masks are made for
          mp.sites = c->d_S;
          mp.spitch = c->nwh + 1;
          rc = launch_plong(c, p, mp);
        }
        else
          rc = wide ? launch_pcross_mw<4, kRowsPerWavePacked>(c, p, mp, tt_for(tcount))
                    : launch_pcross_mw<4, 1>(c, p, mp, tt_for(tcount));
      } else if (turns > 1) {
        rc = launch_prow_r<bh::MODE_CROSS, kRowsPerWaveLong>(c, p, 0);
      } else {
        rc = launch_prow_mode<bh::MODE_CROSS>(c, p, tt_for(tcount));
      }
    } else {
      bh::RowParams p = base_params(c);
      p.src = static_cast<const uint8_t *>(d_pool);
      p.out = d_out_bytes;
      p.idx = d_idx;
      p.gebv = d_gebv;
      p.Z = c->d_Z[buf];
      p.n_pool = (int32_t)pool_rows;
      p.n_out = (int32_t)n_out;
      p.n_rows = (int32_t)rows;
      p.rng_shared = shared;
      p.t0 = 0;
      p.tcount = tcount;
      p.reward = reward_in_launch ? d_reward : nullptr;
      p.reward_next = d_reward_next;
      p.reward_pitch = reward_pitch;
      p.n_envs = (int32_t)n_envs;
      if (run_ahead) {
        if (mw == 1)
          rc = wide ? launch_cross_mw<1, kRowsPerWaveCross>(c, p, mp, tt_for(tcount))
                    : launch_cross_mw<1, 1>(c, p, mp, tt_for(tcount));
        else
          rc = wide ? launch_cross_mw<4, kRowsPerWaveCross>(c, p, mp, tt_for(tcount))
                    : launch_cross_mw<4, 1>(c, p, mp, tt_for(tcount));
      } else {
        rc = launch_row_mode<bh::MODE_CROSS>(c, p, tt_for(tcount));
      }
    }
    if (rc) return rc;
    if (int rc2 = flush_founders(c)) return rc2;  // a deferred founder draw this kind of launch could not carry
    if (run_ahead) {
      c->pf.valid = true;
      c->pf.buf = 1 - buf;
      c->pf.key0 = nk.x;
      c->pf.key1 = nk.y;
      c->pf.B = Bn;
      c->pf.dh = dhn;
      c->pf.off = kwn.off;
      c->pf.total = kwn.total;
    }
    if (mutate)
      if (int rc2 = mutate_rows(c, sim_key[0], sim_key[1], nkeys, kw, B, dh, shared, rows, n_out, d_out_packed, d_out_bytes))
        return rc2;
    if (d_gebv && !fused) {
      if (packed)
        rc = mfma_after ? gebv_bits(c, d_out_packed, rows, nullptr, rows, d_gebv)
                        : packed_gebv_chunks(c, d_out_packed, rows, d_gebv, 0);
      else
        rc = gebv_chunks(c, d_out_bytes, rows, d_gebv, 0);
      if (rc) return rc;
    }
    // the reward of vec_env.py:97 where the GEBV came from a pass of its own: a reduction kernel behind it
    if (d_reward && !reward_in_launch) {
      hipLaunchKernelGGL(bh::reward_max_kernel, dim3((unsigned)n_envs), dim3(256), 0, c->stream, d_gebv, n_out * c->T,
                         d_reward, reward_pitch);
      BH_HIP(hipGetLastError());
    }
    c->primed_reward = d_reward_next;
    c->primed_envs = d_reward_next ? n_envs : 0;
    c->primed_pitch = reward_pitch;
  } else if (d_reward_next && n_envs > 0) {
    hipLaunchKernelGGL(bh::prime_reward_kernel, dim3(1), dim3(256), 0, c->stream, d_reward_next, (int)n_envs, reward_pitch);
    BH_HIP(hipGetLastError());
    c->primed_reward = d_reward_next;
    c->primed_envs = n_envs;
    c->primed_pitch = reward_pitch;
  }
  sim_key[0] = nk.x;
  sim_key[1] = nk.y;
  return BH_OK;
}

int bh_cross(bh_ctx *c, const uint8_t *d_pool, int64_t n_envs, int64_t pool_rows, const int32_t *d_idx,
             int64_t n_out, uint32_t sim_key[2], int rng_mode, uint8_t *d_out, float *d_gebv, float *d_reward,
             float *d_reward_next, int64_t reward_pitch) {
  return cross_impl(c, "bh_cross", false, d_pool, n_envs, pool_rows, d_idx, n_out, sim_key, rng_mode, d_out, nullptr,
                    d_gebv, d_reward, d_reward_next, reward_pitch);
}

int bh_cross_packed(bh_ctx *c, const uint16_t *d_pool, int64_t n_envs, int64_t pool_rows, const int32_t *d_idx,
                    int64_t n_out, uint32_t sim_key[2], int rng_mode, uint16_t *d_out_packed, uint8_t *d_out_bytes,
                    float *d_gebv, float *d_reward, float *d_reward_next, int64_t reward_pitch) {
  return cross_impl(c, "bh_cross_packed", true, d_pool, n_envs, pool_rows, d_idx, n_out, sim_key, rng_mode,
                    d_out_bytes, d_out_packed, d_gebv, d_reward, d_reward_next, reward_pitch);
}

int bh_hint_next_cross(bh_ctx *c, int64_t n_envs, int64_t n_out, int rng_mode) {
  if (!c) return fail(BH_ERR_BAD_ARG, "null context");
  if (n_envs < 0 || n_out < 0 || rng_mode < 0 || rng_mode > (BH_RNG_INDEPENDENT | BH_RNG_DOUBLE_HAPLOID))
    return fail(BH_ERR_BAD_ARG, "bh_hint_next_cross: bad arguments");
  c->hint.valid = true;
  c->hint.n_envs = n_envs;
  c->hint.n_out = n_out;
  c->hint.rng_mode = rng_mode;
  return BH_OK;
}

int bh_step_bind(bh_ctx *c, int64_t n_envs, int64_t n_out, int64_t depth, void *const *d_state, void *const *d_obs,
                 void *const *d_gebv, int64_t n_rewards, void *const *d_reward, int64_t reward_pitch) {
  if (int rc = check_ctx(c)) return rc;
  if (depth == 0) {  // unbind
    c->ring = {};
    return BH_OK;
  }
  if (n_envs <= 0 || n_out <= 0 || depth < 2 || depth > 4096 || n_rewards < 0 || n_rewards > 4096 || reward_pitch < 1 ||
      !d_state || !d_gebv || (n_rewards > 0 && !d_reward))
    return fail(BH_ERR_BAD_ARG, "bh_step_bind: need n_envs, n_out > 0, 2 <= depth <= 4096, reward_pitch >= 1 and the block tables");
  for (int64_t s = 0; s < depth; ++s)
    if (!d_state[s] || !d_gebv[s] || (d_obs && !d_obs[s]))
      return fail(BH_ERR_BAD_ARG, "bh_step_bind: slot %lld has a null block", (long long)s);
  for (int64_t s = 0; s < n_rewards; ++s)
    if (!d_reward[s]) return fail(BH_ERR_BAD_ARG, "bh_step_bind: reward block %lld is null", (long long)s);
  c->ring.n_envs = n_envs;
  c->ring.n_out = n_out;
  c->ring.depth = depth;
  c->ring.n_rewards = n_rewards;
  c->ring.reward_pitch = reward_pitch;
  c->ring.state.assign(d_state, d_state + depth);
  c->ring.gebv.assign(d_gebv, d_gebv + depth);
  c->ring.reward.assign(d_reward, d_reward + n_rewards);
  if (d_obs)
    c->ring.obs.assign(d_obs, d_obs + depth);
  else
    c->ring.obs.assign((size_t)depth, nullptr);
  return BH_OK;
}

int bh_step(bh_ctx *c, int64_t slot, int64_t reward_slot, int64_t prime_slot, const uint16_t *d_pool, int64_t pool_rows,
            const int32_t *d_actions, uint32_t sim_key[2], int flags) {
  if (!c) return fail(BH_ERR_BAD_ARG, "null context");
  const auto &r = c->ring;
  if (r.depth == 0) return fail(BH_ERR_BAD_ARG, "bh_step: no blocks bound (bh_step_bind)");
  if (slot < 0 || slot >= r.depth || reward_slot >= r.n_rewards || prime_slot >= r.n_rewards ||
      (prime_slot >= 0 && prime_slot == reward_slot))
    return fail(BH_ERR_BAD_ARG, "bh_step: slot %lld / reward %lld / prime %lld outside the bound tables (%lld, %lld)",
                (long long)slot, (long long)reward_slot, (long long)prime_slot, (long long)r.depth, (long long)r.n_rewards);
  if ((flags & BH_STEP_OBSERVATION) && !r.obs[(size_t)slot])
    return fail(BH_ERR_BAD_ARG, "bh_step: the blocks were bound without observations");
  return cross_impl(c, "bh_step", true, d_pool, r.n_envs, pool_rows, d_actions, r.n_out, sim_key, flags & 3,
                    (flags & BH_STEP_OBSERVATION) ? static_cast<uint8_t *>(r.obs[(size_t)slot]) : nullptr,
                    static_cast<uint16_t *>(r.state[(size_t)slot]), static_cast<float *>(r.gebv[(size_t)slot]),
                    reward_slot >= 0 ? static_cast<float *>(r.reward[(size_t)reward_slot]) : nullptr,
                    prime_slot >= 0 ? static_cast<float *>(r.reward[(size_t)prime_slot]) : nullptr, r.reward_pitch);
}

int64_t bh_packed_pitch(const bh_ctx *c) { return c ? (int64_t)16 * c->pw : 0; }

int bh_pack(bh_ctx *c, const uint8_t *d_rows, int64_t n_rows, uint16_t *d_packed) {
  if (int rc = check_ctx(c)) return rc;
  if (n_rows < 0) return fail(BH_ERR_BAD_ARG, "bh_pack: negative size");
  if (n_rows == 0) return BH_OK;
  if (!d_rows || !d_packed) return fail(BH_ERR_BAD_ARG, "bh_pack: null device pointer");
  if (n_rows > 0x3FFFFFFF) return fail(BH_ERR_UNSUPPORTED, "bh_pack: too many rows");
  bh::PackParams p{};
  p.src = d_rows;
  p.dst = d_packed;
  p.pitch = c->pitch;
  p.nvec = c->nvec;
  p.nvp = 8 * c->pw;
  p.n_rows = (int32_t)n_rows;
  return launch_pack(c, p);
}

static int check_gather_args(bh_ctx *c, const char *what, const uint16_t *d_src, int64_t src_rows, const int32_t *d_idx,
                             int64_t n_rows, float *d_gebv) {
  if (n_rows < 0 || src_rows < 0) return fail(BH_ERR_BAD_ARG, "%s: negative size", what);
  if (n_rows == 0) return BH_OK;
  if (!d_src) return fail(BH_ERR_BAD_ARG, "%s: null device pointer", what);
  if (src_rows == 0) return fail(BH_ERR_BAD_ARG, "%s: empty source", what);
  if (!d_idx && n_rows > src_rows) return fail(BH_ERR_BAD_ARG, "%s: more rows than the source holds", what);
  if (n_rows > 0x3FFFFFFF || src_rows > 0x7FFFFFFF) return fail(BH_ERR_UNSUPPORTED, "%s: too many rows", what);
  if (d_gebv && c->T == 0) return fail(BH_ERR_BAD_ARG, "%s: GEBV requested but the context has no traits", what);
  return BH_OK;
}

int bh_gather_packed(bh_ctx *c, const uint16_t *d_src, int64_t src_rows, const int32_t *d_idx, int64_t n_rows,
                     uint16_t *d_out_packed, uint8_t *d_out_bytes, float *d_gebv) {
  if (int rc = check_ctx(c)) return rc;
  if (int rc = flush_founders(c)) return rc;
  if (int rc = check_gather_args(c, "bh_gather_packed", d_src, src_rows, d_idx, n_rows, d_gebv)) return rc;
  if (n_rows == 0) return BH_OK;
  const GatherPlan g = plan_gather(c, d_src, src_rows, d_idx, n_rows, d_out_packed, d_out_bytes, d_gebv);
  return run_gather(c, g, d_src, src_rows, d_idx, n_rows, d_gebv);
}

int bh_gather_packed_deferred(bh_ctx *c, const uint16_t *d_src, int64_t src_rows, const int32_t *d_idx, int64_t n_rows,
                              uint16_t *d_out_packed, uint8_t *d_out_bytes, float *d_gebv) {
  if (int rc = check_ctx(c)) return rc;
  // a founder draw still pending may be what writes d_idx: then it is made now, in stream order; any other stays
  // pending and rides with this gather (VecBreedGym's horizon step: the founders of the reset after the next one)
  if (c->founders_pending && static_cast<const int32_t *>(c->fp_pending.out) == d_idx)
    if (int rc = flush_founders_draw(c)) return rc;
  if (int rc = flush_gather(c)) return rc;  // an earlier gather asked for ahead
  if (int rc = check_gather_args(c, "bh_gather_packed_deferred", d_src, src_rows, d_idx, n_rows, d_gebv)) return rc;
  if (n_rows == 0) return BH_OK;
  c->gather_ahead = {d_src, src_rows, d_idx, n_rows, d_out_packed, d_out_bytes, d_gebv, true};
  return BH_OK;
}

int bh_set_mask_prefetch(bh_ctx *c, int enable) {
  if (!c) return fail(BH_ERR_BAD_ARG, "null context");
  c->prefetch = enable != 0;
  return BH_OK;
}

int bh_gather(bh_ctx *c, const uint8_t *d_src, int64_t src_rows, const int32_t *d_idx, int64_t n_rows,
              uint8_t *d_out, float *d_gebv) {
  if (int rc = check_ctx(c)) return rc;
  if (int rc = flush_founders(c)) return rc;
  if (n_rows < 0 || src_rows < 0) return fail(BH_ERR_BAD_ARG, "bh_gather: negative size");
  if (n_rows == 0) return BH_OK;
  if (!d_src || !d_idx || !d_out) return fail(BH_ERR_BAD_ARG, "bh_gather: null device pointer");
  if (src_rows == 0) return fail(BH_ERR_BAD_ARG, "bh_gather: empty source");
  if (n_rows > 0x3FFFFFFF || src_rows > 0x7FFFFFFF) return fail(BH_ERR_UNSUPPORTED, "bh_gather: too many rows");
  if (d_gebv && c->T == 0) return fail(BH_ERR_BAD_ARG, "bh_gather: GEBV requested but the context has no traits");
  bh::RowParams p = base_params(c);
  p.src = d_src;
  p.out = d_out;
  p.idx = d_idx;
  p.gebv = d_gebv;
  p.n_pool = (int32_t)src_rows;
  p.n_rows = (int32_t)n_rows;
  // (large gathers: the rows first, then the one-pass matrix-core GEBV over them -- 2048 x 100k x 1: 145 + 74 us against
  // 256 us with the f64 GEBV in the gather's stream, profiles/r05_scan_ops.txt)
  const bool fused = d_gebv && c->T <= kMaxFusedTraits && !use_mfma_gebv(c, n_rows) &&
                     !(use_bits_gebv(c, n_rows) && n_rows * c->M >= kGebvViaPlanesMin);
  p.t0 = 0;
  p.tcount = fused ? (int32_t)c->T : 0;
  if (int rc = launch_row_mode<bh::MODE_GATHER>(c, p, tt_for(p.tcount))) return rc;
  if (d_gebv && !fused) return gebv_chunks(c, d_out, n_rows, d_gebv, 0);
  return BH_OK;
}

int bh_gebv(bh_ctx *c, const uint8_t *d_pop, int64_t rows, float *d_gebv) {
  if (int rc = check_ctx(c)) return rc;
  if (int rc = flush_founders(c)) return rc;
  if (rows < 0) return fail(BH_ERR_BAD_ARG, "bh_gebv: negative size");
  if (rows == 0 || c->T == 0) return BH_OK;
  if (!d_pop || !d_gebv) return fail(BH_ERR_BAD_ARG, "bh_gebv: null device pointer");
  if (rows > 0x3FFFFFFF) return fail(BH_ERR_UNSUPPORTED, "bh_gebv: too many rows");
  return gebv_chunks(c, d_pop, rows, d_gebv, 0);
}

int bh_reward_max(bh_ctx *c, const float *d_gebv, int64_t n_envs, int64_t per_env, float *d_reward) {
  if (int rc = check_ctx(c)) return rc;
  if (n_envs < 0 || per_env < 0) return fail(BH_ERR_BAD_ARG, "bh_reward_max: negative size");
  if (n_envs == 0) return BH_OK;
  if (!d_gebv || !d_reward) return fail(BH_ERR_BAD_ARG, "bh_reward_max: null device pointer");
  hipLaunchKernelGGL(bh::reward_max_kernel, dim3((unsigned)n_envs), dim3(256), 0, c->stream, d_gebv, per_env,
                     d_reward, (int64_t)1);
  BH_HIP(hipGetLastError());
  return BH_OK;
}

}  // extern "C"

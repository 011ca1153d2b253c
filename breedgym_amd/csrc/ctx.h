// Host-side state shared by the translation units of libbreedhip.so:
//   breedhip.hip   context, memory, settings, host key plumbing
//   cross.hip      bh_cross*, bh_gather*, bh_pack, bh_gebv (dispatch), bh_reward_max  -- the step/reset launches
//   gebv.hip       the matrix-core GEBV passes, bh_optimal_haploid
//   select.hip     bh_topk, bh_selection_actions, bh_founders_device/_deferred, the radix sort in HBM, bh_winner_layout
// One kernel header belongs to exactly one of them (non-template __global__ functions are defined in headers).
#pragma once
#include "breedhip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <numeric>
#include <vector>

#include "founders.h"
#include "threefry.h"

namespace bhh {  // host helpers, defined in breedhip.hip unless noted

int fail(int code, const char *fmt, ...);

#define BH_HIP(call)                                                                          \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess)                                                                     \
      return bhh::fail(e_ == hipErrorOutOfMemory ? BH_ERR_OOM : BH_ERR_HIP, "%s failed: %s", \
                       #call, hipGetErrorString(e_));                                         \
  } while (0)

constexpr int kMaxFusedTraits = 8;
// digit layout of the int8 matrix-core GEBV (gebv_mfma.h, gebv_bits.h)
constexpr int kGebvDigits = 6;
constexpr int kGebvCols = 64;  // two 32-column MFMA tiles; T * 6 <= 64  =>  T <= 10

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

}  // namespace bhh

struct bh_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  int64_t M = 0, T = 0, pitch = 0;
  int32_t half = 0, nwh = 0, nw = 0, nvec = 0;
  // 16-byte words from one bit-plane row (and one row of crossover masks) to the next: nw, and whole 128-byte lines for
  // rows beyond 4 KB (bh_create) -- what bh_packed_pitch() hands out.  Words nw .. pw - 1 are padding like the bits past
  // marker M: zero in every row the library writes, never looked at as markers.
  int32_t pw = 0;
  int32_t zw = 0;  // ... and from one row of crossover masks to the next
  int64_t effPitch = 0;
  uint32_t *d_thrA = nullptr, *d_thrB = nullptr;  // 32-bit thresholds (thr23 << 9)
  uint32_t *d_ctrB = nullptr;                     // second Threefry counter per pair
  uint64_t *d_alwA = nullptr, *d_alwB = nullptr;  // site words of markers with r >= 1, or null
  int64_t nthr = 0;
  double *d_eff = nullptr;     // (T, effPitch)
  double *d_effsum = nullptr;  // (effPitch) sum over traits (block values of the optimal haploid)
  // crossover-mask buffers: the launch that streams generation g from d_Z[cur] also makes the
  // masks of generation g+1 in d_Z[1 - cur] (cross_kernel's two roles)
  uint8_t *d_Z[2] = {nullptr, nullptr};
  int64_t z_cap[2] = {0, 0};  // bytes
  char *d_spare = nullptr;    // 16 KB nobody reads (PackedParams::spare)
  uint64_t *d_S = nullptr;    // site words of the stream-per-lane mask role (mask_stream.h)
  int64_t s_cap = 0;          // bytes
  float mutation = 0.0f;      // bh_set_mutation: chromax's mutation probability (0: off, the hot path)
  uint32_t mut_thr = 0;       // (bits >> 9) < mut_thr  <=>  uniform_f32(bits) < mutation
  uint64_t *d_F = nullptr;    // flip words of the keyed gametes (mutation.h)
  int64_t f_cap = 0;          // bytes
  bool prefetch = true;
  int layout = bh::LAYOUT_ORIGINAL;  // bh_set_threefry_layout
  // sharding windows: this context holds envs [env_offset, env_offset + n_envs) of envs_total, or crosses
  // [cross_offset, ..) of crosses_total of one batch, and draws the keys the whole would draw (0 = no window)
  int64_t env_offset = 0, envs_total = 0, cross_offset = 0, crosses_total = 0;
  struct {
    bool valid = false;
    int buf = 0;
    uint32_t key0 = 0, key1 = 0;
    int64_t B = 0;
    bool dh = false;
    int64_t off = 0, total = 0;
  } pf;
  struct {  // bh_hint_next_cross: what the NEXT bh_cross* call will be, where it differs from this one
    bool valid = false;
    int64_t n_envs = 0, n_out = 0;
    int rng_mode = 0;
  } hint;
  int last_buf = 1;
  // fixed-point marker effects for the int8-MFMA GEBV: one pack per group of <= 10 traits
  std::vector<void *> d_Bpack;  // bh::i32x4 rows (gebv_mfma.h)
  std::vector<void *> d_Bbits;  // the same digits in the operand order of gebv_bits_kernel
  int32_t n_mc = 0;                  // 512-marker macro-chunks per row
  double *d_scale = nullptr;  // (T) 2^-s_t
  uint4 *d_dig16 = nullptr;   // (T, 3, nvec + 1) int16 digit planes, the last block of a plane zero, of the fixed-point effects (packed_kernels.h)
  int32_t n_chunks = 0;
  int32_t *d_partial = nullptr;  // gebv_mfma_kernel's int32 slabs
  int64_t partial_cap = 0;       // int32 words
  void *d_gsum = nullptr;  // gebv_bits.h, GebvOut: the final cells, then one set of cells per slice group
  int64_t gsum_cap = 0;    // bytes
  uint8_t *d_pos8 = nullptr;    // (nvec) bit i: the first trait's effect of marker 8 v + i is positive (index_kernels.h)
  void *d_ohp = nullptr;        // bh_optimal_haploid: the haploids as bit-plane rows, then the block of every marker
  int64_t ohp_cap = 0;
  void *d_pool_gebv = nullptr;  // gebv_bits: the GEBV of a pool's rows, on its way through an index (gebv.hip)
  int64_t pool_gebv_cap = 0;
  // top-k scratch
  float *d_cand_s[2] = {nullptr, nullptr};
  int32_t *d_cand_i[2] = {nullptr, nullptr};
  int64_t cand_cap = 0;
  // the radix sort in HBM (select.hip): keys and values twice, tile histograms, per-round subkeys; top-k winners
  void *d_sort = nullptr, *d_best = nullptr;
  int64_t sort_cap = 0, best_cap = 0;
  // a founder draw asked for ahead (bh_founders_deferred): rides on the next two-role launch, or gets its own
  bh::FoundersParams fp_pending{};
  bool founders_pending = false;
  // a gather asked for ahead (bh_gather_packed_deferred): rides on the next two-role launch of bit-plane rows with its
  // GEBV in the stream (packed_cross_kernel, gp), or is made by whatever comes first (flush_founders)
  struct {
    const uint16_t *src = nullptr;
    int64_t src_rows = 0;
    const int32_t *idx = nullptr;
    int64_t n_rows = 0;
    uint16_t *outp = nullptr;
    uint8_t *outb = nullptr;
    float *gebv = nullptr;
    bool pending = false;
  } gather_ahead;
  // reward chaining (bh_cross*): the buffer the last launch left holding -inf, ready to take atomic maxima
  const float *primed_reward = nullptr;
  int64_t primed_envs = 0, primed_pitch = 0;
  // bh_step_bind: the caller's ring of output blocks for bh_step
  struct {
    int64_t n_envs = 0, n_out = 0, depth = 0, n_rewards = 0, reward_pitch = 1;
    std::vector<void *> state, obs, gebv, reward;
  } ring;
  bool big_lds_set = false;  // topk_chunk_kernel's dynamic-LDS limit raised (select.hip)
  std::vector<const void *> big_lds_kernels;  // kernel instances whose dynamic-LDS limit has been raised (allow_big_lds)
  bool select_lds_set = false;
  bool sparse_lds_set = false;
  unsigned bits_lds_set = 0;  // gebv_bits_kernel instances whose dynamic LDS limit has been raised
  // device-side error word: host memory the kernels can write (a hand-off wait that ran out, row_kernels.h)
  uint32_t *h_err = nullptr, *d_err = nullptr;
};

namespace bhh {

int check_ctx(const bh_ctx *c);
// grow a device scratch buffer (waits for the stream first; contents are lost)
int ensure_scratch(bh_ctx *c, void **buf, int64_t *cap, int64_t bytes);

// gebv.hip: GEBV passes on the matrix cores, all traits
int gebv_mfma(bh_ctx *c, const uint8_t *d_pop, int64_t rows, float *d_gebv);
int gebv_bits(bh_ctx *c, const uint16_t *d_rows, int64_t n_src, const int32_t *d_idx, int64_t rows, float *d_gebv);
int gebv_bytes(bh_ctx *c, const uint8_t *d_pop, int64_t rows, float *d_gebv);

// select.hip: a deferred founder draw that found no launch to ride on is made now, in stream order -- and behind it a
// gather asked for ahead that found none (cross.hip)
int flush_founders(bh_ctx *c);
int flush_founders_draw(bh_ctx *c);  // the draw alone
int flush_gather(bh_ctx *c);

}  // namespace bhh

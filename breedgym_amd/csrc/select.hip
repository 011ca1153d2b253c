// libbreedhip.so -- ranking and draws: bh_topk, bh_selection_actions, bh_founders_device / _deferred.
//
// Which route a call takes:
//   bh_topk               n <= 16384: bitonic sort in LDS (topk_chunk_kernel);  n > 16384, k <= 16384: radix select
//                         + sort of the winners (topk_select_kernel);  k > 16384: radix sort in HBM (sort_kernels.h)
//   bh_founders_device    germplasm <= 8192: one workgroup per env in LDS (founders.h; <= 1024 rows may ride on the
//                         horizon launch);  above: JAX's shuffle rounds as radix sorts in HBM, envs side by side
//   bh_selection_actions  n <= 8192 and k(k-1)/2 <= 8192: one workgroup per env in LDS (selection_actions_kernel);
//                         above: bh_topk's routes for the k best, radix-sort shuffle of the pair ranks, un-ranking
//   bh_pair_actions       bh_topk's routes for the n best of an env's pair scores, then one workgroup per env
//                         (softmax, offspring counts, repeat: pair_score_actions_kernel)
#include "ctx.h"
#include "select_kernels.h"
#include "shard_kernels.h"
#include "shuffle_sparse.h"
#include "sort_kernels.h"

using namespace bhh;

namespace {

constexpr int kTopkChunk = 16384;  // candidates one block sorts in LDS (128 KiB of keys)

int launch_founders(bh_ctx *c, const bh::FoundersParams &p) {
  const size_t lds = (size_t)p.P * 16;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(bh::founders_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return fail(BH_ERR_HIP, "hipFuncSetAttribute(founders LDS): %s", hipGetErrorString(e));
  }
  const int threads = (int)std::min<int64_t>(1024, std::max<int64_t>(64, p.P / 2));
  hipLaunchKernelGGL(bh::founders_kernel, dim3((unsigned)p.E), dim3(threads), lds, c->stream, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(BH_ERR_HIP, "founders_kernel launch: %s", hipGetErrorString(e));
  return BH_OK;
}


// ---- the radix sort in HBM (sort_kernels.h) -----------------------------------------------------------------
struct SortBufs {
  uint2 *A, *B;  // (key, value) elements, twice
  uint32_t *hist, *tot, *sub;
  int64_t stride;  // elements per segment in the four arrays
  int32_t nb;      // tiles per segment
};

// segments one batch sorts side by side: <= 2^25 elements (0.5 GB of scratch), <= 65535 (grid.y)
int64_t sort_batch(int64_t segs, int64_t N) {
  const int64_t stride = round_up(N, 64);
  return std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(segs, 65535), ((int64_t)1 << 25) / stride));
}

int sort_layout(bh_ctx *c, int64_t segs, int64_t N, int64_t rounds, SortBufs *b) {
  b->stride = round_up(N, 64);
  b->nb = (int32_t)((N + bh::kSortTile - 1) / bh::kSortTile);
  const int64_t arr = segs * b->stride * 8, hist = segs * 256 * b->nb * 4, tot = segs * 256 * 4,
                sub = round_up(segs * std::max<int64_t>(rounds, 1) * 8, 256);
  void *base = c->d_sort;
  if (int rc = ensure_scratch(c, &base, &c->sort_cap, 2 * arr + hist + tot + sub)) {
    c->d_sort = base;
    return rc;
  }
  c->d_sort = base;
  char *q = static_cast<char *>(base);
  b->A = reinterpret_cast<uint2 *>(q);
  b->B = reinterpret_cast<uint2 *>(q + arr);
  b->hist = reinterpret_cast<uint32_t *>(q + 2 * arr);
  b->tot = reinterpret_cast<uint32_t *>(q + 2 * arr + hist);
  b->sub = reinterpret_cast<uint32_t *>(q + 2 * arr + hist + tot);
  return BH_OK;
}

// one stable pass on 8 bits of the key: in -> out (tiles of `nb` per segment; hist / tot as in `b`)
int radix_pass(bh_ctx *c, const SortBufs &b, const uint2 *in, uint2 *out, int64_t segs, int64_t N, int shift) {
  bh::SortParams p{};
  p.in = in;
  p.out = out;
  p.hist = b.hist;
  p.bin_total = b.tot;
  p.seg_stride = b.stride;
  p.N = (int32_t)N;
  p.nb = (int32_t)((N + bh::kSortTile - 1) / bh::kSortTile);
  p.shift = shift;
  const dim3 tiles((unsigned)p.nb, (unsigned)segs);
  hipLaunchKernelGGL(bh::sort_hist_kernel, tiles, dim3(bh::kSortThreads), 0, c->stream, p);
  hipLaunchKernelGGL(bh::sort_scan_kernel, dim3(256, (unsigned)segs), dim3(256), 0, c->stream, p);
  hipLaunchKernelGGL(bh::sort_scatter_kernel, tiles, dim3(bh::kSortThreads), 0, c->stream, p);
  BH_HIP(hipGetLastError());
  return BH_OK;
}

// the first N elements of every segment of `x` sorted by key, stable, back into `x` (`y`: the other array of `b`)
int radix_sort(bh_ctx *c, const SortBufs &b, uint2 *x, uint2 *y, int64_t segs, int64_t N) {
  for (int pass = 0; pass < 4; ++pass)
    if (int rc = radix_pass(c, b, (pass & 1) ? y : x, (pass & 1) ? x : y, segs, N, 8 * pass)) return rc;
  return BH_OK;
}

int shuffle_rounds_for(int64_t n) {  // jax/_src/random.py `_shuffle`
  return (int)std::ceil(3.0 * std::log((double)std::max<int64_t>(1, n)) / std::log(4294967295.0));
}

// jax.random.permutation(key_e, N)[:take] for envs [env_first, env_first + segs) of E_total, key_e = split(env_key,
// E_total + 1)[1 + e], side by side as the values of the first `take` elements of every segment of *result (b.A,
// segment stride b.stride): every round a full stable radix sort of all N elements.  The route of draws that want
// (nearly) all of a long list; everything else goes through shuffle_front_sparse.
int shuffle_segments(bh_ctx *c, const uint32_t env_key[2], int64_t E_total, int64_t env_first, int64_t segs, int64_t N,
                     int rounds, const SortBufs &b, const uint2 **result) {
  *result = b.A;
  bh::ShuffleKeyParams kp{};
  kp.subkeys = b.sub;
  kp.key0 = env_key[0];
  kp.key1 = env_key[1];
  kp.segs = (int32_t)segs;
  kp.rounds = rounds;
  kp.layout = c->layout;
  kp.E_total = (int32_t)E_total;
  kp.env_first = (int32_t)env_first;
  if (rounds > 0) {
    hipLaunchKernelGGL(bh::shuffle_subkeys_kernel, dim3((unsigned)((segs + 63) / 64)), dim3(64), 0, c->stream, kp);
    BH_HIP(hipGetLastError());
  }
  bh::ShuffleGenParams gp{};
  gp.subkeys = b.sub;
  gp.kv = b.A;
  gp.seg_stride = b.stride;
  gp.N = (int32_t)N;
  gp.rounds = rounds;
  gp.layout = c->layout;
  const int64_t per_seg = c->layout == bh::LAYOUT_PARTITIONABLE ? N : (N + 1) / 2;
  const dim3 grid((unsigned)((per_seg + 255) / 256), (unsigned)segs);
  for (int rd = 0; rd < std::max(rounds, 1); ++rd) {  // N = 1: no round, the identity
    gp.round = rd;
    gp.init = rd == 0;
    hipLaunchKernelGGL(bh::shuffle_gen_kernel, grid, dim3(256), 0, c->stream, gp);
    BH_HIP(hipGetLastError());
    if (rounds == 0) break;
    if (int rc = radix_sort(c, b, b.A, b.B, segs, N)) return rc;
  }
  return BH_OK;
}

// ---- the wanted-only shuffle (shuffle_sparse.h) -------------------------------------------------------------
// Which draws take it: lists beyond one workgroup's LDS, up to 2^25 elements (a bucket then holds 512 on average),
// and -- above 2^18 elements, where a bucket is no longer a handful -- only when a sixteenth of the list or less is
// wanted (a wave per wanted element looks through its whole bucket).
bool sparse_route(int64_t N, int64_t take) {
  return N > 8192 && N <= ((int64_t)1 << 25) && take >= 1 && (N <= ((int64_t)1 << 18) || take * 16 <= N);
}

struct SparseGeom {
  int32_t chunk_items, nchunk, reduce_batch, nchunk_pad;  // sparse_reduce_kernel's BATCH; rows incl. its padding
  int64_t take_stride, list_stride, per_seg_bytes;
};

SparseGeom sparse_geom(int64_t segs, int64_t N, int64_t take, int layout) {
  SparseGeom g{};
  const int64_t items = bh::sparse_items(layout, (int)N);
  // ONE round of workgroups on the 256 CUs (a workgroup's 128 KB of counters leave room for one per CU): every chunk
  // is a row of 65,536 16-bit counts that sparse_hist writes, sparse_reduce reads twice and rewrites and sparse_gather
  // reads again -- at 21.5 M elements those passes were 320 MB per round with 512 chunks (two rounds of workgroups, each
  // within the 65,535 positions a 16-bit count cannot wrap on), more than the sweeps' own time.  Longer chunks could wrap
  // a count only with more than 65,535 equal top halves in one of them, which sparse_hist then sees and reports
  // (kSpErrBucket, as sparse_select does for a bucket over its limit).
  constexpr int64_t kTargetChunks = 256;  // (384: 0.50 ms, 512: 0.49, 256: 0.41 at 21.5 M elements -- profiles/r05_draws.txt)
  const int64_t max_items = layout == bh::LAYOUT_PARTITIONABLE ? 4 * bh::kSpMaxPositions : 2 * bh::kSpMaxPositions;
  const int64_t target = std::max<int64_t>(1, kTargetChunks / std::max<int64_t>(1, segs));
  g.chunk_items = (int32_t)std::min(max_items, std::max<int64_t>(2048, (items + target - 1) / target));
  g.nchunk = (int32_t)((items + g.chunk_items - 1) / g.chunk_items);
  const int rows = (g.nchunk + 15) / 16;  // rows per wave of sparse_reduce, asked for reduce_batch at a time
  g.reduce_batch = rows <= 1 ? 1 : rows <= 4 ? 4 : rows <= 8 ? 8 : rows <= 12 ? 12 : rows <= 16 ? 16
                   : (round_up(rows, 16) <= round_up(rows, 12) ? 16 : 12);
  g.nchunk_pad = 16 * (int32_t)round_up(rows, g.reduce_batch);
  g.take_stride = round_up(take, 64);
  g.list_stride = round_up(N, 64);
  g.per_seg_bytes = (int64_t)g.nchunk_pad * bh::kSpBuckets * 2 + (int64_t)(2 * bh::kSpBuckets + bh::kSpGroups) * 4 + bh::kSpBuckets / 8 +
                    5 * g.take_stride * 4 + g.list_stride * 8 + g.take_stride * 8 + 16 * 256;
  return g;
}

// segments one batch draws side by side: about 1 GB of scratch at most
int64_t sparse_batch(int64_t segs, int64_t N, int64_t take, int layout) {
  const SparseGeom g = sparse_geom(1, N, take, layout);
  return std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(segs, 65535), ((int64_t)1 << 30) / g.per_seg_bytes));
}

// permutation(key_e, N)[:take] as shuffle_segments, by the wanted-only route: *result (segs, *stride) elements whose
// .y are the values.  Nothing is synchronised; a bucket over the limit leaves a code in the device error word.
int shuffle_front_sparse(bh_ctx *c, const uint32_t env_key[2], int64_t E_total, int64_t env_first, int64_t segs, int64_t N,
                         int rounds, int64_t take, const uint2 **result, int64_t *stride) {
  const SparseGeom g = sparse_geom(segs, N, take, c->layout);
  const int64_t sub_bytes = round_up(segs * std::max(rounds, 1) * 8, 256);
  void *base = c->d_sort;
  const int rc0 = ensure_scratch(c, &base, &c->sort_cap, segs * g.per_seg_bytes + sub_bytes + 4096);
  c->d_sort = base;
  if (rc0) return rc0;
  char *q = static_cast<char *>(base);
  auto carve = [&](int64_t bytes) {
    char *r = q;
    q += round_up(bytes, 256);
    return r;
  };
  bh::SparseParams p{};
  uint32_t *sub = reinterpret_cast<uint32_t *>(carve(sub_bytes));
  p.subkeys = sub;
  p.partial = reinterpret_cast<uint16_t *>(carve(segs * (int64_t)g.nchunk_pad * bh::kSpBuckets * 2));
  p.list = reinterpret_cast<uint2 *>(carve(segs * g.list_stride * 8));
  p.out = reinterpret_cast<uint2 *>(carve(segs * g.take_stride * 8));
  p.total = reinterpret_cast<uint32_t *>(carve(segs * (int64_t)bh::kSpBuckets * 4));
  p.pre16 = reinterpret_cast<uint32_t *>(carve(segs * (int64_t)bh::kSpBuckets * 4));
  p.group_sum = reinterpret_cast<uint32_t *>(carve(segs * (int64_t)bh::kSpGroups * 4));
  p.marks = reinterpret_cast<uint32_t *>(carve(segs * (int64_t)bh::kSpBuckets / 8));
  uint32_t *want[2] = {reinterpret_cast<uint32_t *>(carve(segs * take * 4)), reinterpret_cast<uint32_t *>(carve(segs * take * 4))};
  p.wbucket = reinterpret_cast<uint32_t *>(carve(segs * take * 4));
  p.wrank = reinterpret_cast<uint32_t *>(carve(segs * take * 4));
  p.wstart = reinterpret_cast<uint32_t *>(carve(segs * take * 4));
  p.err = c->d_err;
  p.list_stride = g.list_stride;
  p.out_stride = g.take_stride;
  p.N = (int32_t)N;
  p.take = (int32_t)take;
  p.nchunk = g.nchunk;
  p.nchunk_pad = g.nchunk_pad;
  p.chunk_items = g.chunk_items;
  p.rounds = rounds;
  p.layout = c->layout;

  bh::ShuffleKeyParams kp{};
  kp.subkeys = sub;
  kp.key0 = env_key[0];
  kp.key1 = env_key[1];
  kp.segs = (int32_t)segs;
  kp.rounds = rounds;
  kp.layout = c->layout;
  kp.E_total = (int32_t)E_total;
  kp.env_first = (int32_t)env_first;
  hipLaunchKernelGGL(bh::shuffle_subkeys_kernel, dim3((unsigned)((segs + 63) / 64)), dim3(64), 0, c->stream, kp);
  BH_HIP(hipGetLastError());

  constexpr size_t kHistLds = (size_t)bh::kSpLdsWords * 4, kGatherLds = kHistLds + bh::kSpBuckets / 8;
  if (!c->sparse_lds_set) {
    BH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(bh::sparse_hist_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kHistLds));
    BH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(bh::sparse_gather_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGatherLds));
    c->sparse_lds_set = true;
  }
  const dim3 chunks((unsigned)g.nchunk, (unsigned)segs);
  for (int rd = rounds - 1, cur = 0; rd >= 0; --rd, cur ^= 1) {  // the last round first: its front is what is wanted
    p.round = rd;
    p.want = rd == rounds - 1 ? nullptr : want[cur ^ 1];
    p.next_want = want[cur];
    hipLaunchKernelGGL(bh::sparse_hist_kernel, chunks, dim3(bh::kSpThreads), kHistLds, c->stream, p);
    {
      const dim3 blocks(bh::kSpBuckets / 256, (unsigned)segs);
      if (g.reduce_batch == 1)
        hipLaunchKernelGGL(bh::sparse_reduce_kernel<1>, blocks, dim3(1024), 0, c->stream, p);
      else if (g.reduce_batch == 4)
        hipLaunchKernelGGL(bh::sparse_reduce_kernel<4>, blocks, dim3(1024), 0, c->stream, p);
      else if (g.reduce_batch == 8)
        hipLaunchKernelGGL(bh::sparse_reduce_kernel<8>, blocks, dim3(1024), 0, c->stream, p);
      else if (g.reduce_batch == 16)
        hipLaunchKernelGGL(bh::sparse_reduce_kernel<16>, blocks, dim3(1024), 0, c->stream, p);
      else
        hipLaunchKernelGGL(bh::sparse_reduce_kernel<12>, blocks, dim3(1024), 0, c->stream, p);
    }
    hipLaunchKernelGGL(bh::sparse_map_kernel, dim3((unsigned)((take + 255) / 256), (unsigned)segs), dim3(256), 0, c->stream, p);
    hipLaunchKernelGGL(bh::sparse_gather_kernel, chunks, dim3(bh::kSpThreads), kGatherLds, c->stream, p);
    hipLaunchKernelGGL(bh::sparse_select_kernel, dim3((unsigned)((take + 3) / 4), (unsigned)segs), dim3(256), 0, c->stream, p);
    BH_HIP(hipGetLastError());
  }
  *result = p.out;
  *stride = g.take_stride;
  return BH_OK;
}

// The front of every env's shuffle by whichever route fits: `segs` envs from env_first on, at most *done of them (one
// batch of scratch); *result / *stride as above.
int shuffle_front(bh_ctx *c, const uint32_t env_key[2], int64_t E_total, int64_t env_first, int64_t segs_left, int64_t N,
                  int64_t take, const uint2 **result, int64_t *stride, int64_t *done) {
  const int rounds = shuffle_rounds_for(N);
  if (rounds > 0 && sparse_route(N, take)) {
    *done = std::min(sparse_batch(segs_left, N, take, c->layout), segs_left);
    return shuffle_front_sparse(c, env_key, E_total, env_first, *done, N, rounds, take, result, stride);
  }
  *done = std::min(sort_batch(segs_left, N), segs_left);
  SortBufs b{};
  if (int rc = sort_layout(c, *done, N, rounds, &b)) return rc;
  *stride = b.stride;
  return shuffle_segments(c, env_key, E_total, env_first, *done, N, rounds, b, result);
}


// winners from which a long row (more than one workgroup sorts whole) goes through the sort, whose four passes over a
// row of 16 K ... 1 M scores take 83 ... 103 us whatever k: the routes below take 85 us at 512 winners of 40,000 scores,
// 135 at 1,024, 179 at 1,025 (levels of 16384-candidate chunk sorts), 177 for 6,553 of 65,536 (radix select, one
// workgroup per row); tools/time_topk.py
constexpr int64_t kTopkSortMinWinners = 513;
// top-k of every row through the radix sort in HBM: the whole row ascending by ~orderable(score) (stable: ties to the
// lower index), and its first k
static int topk_by_sort(bh_ctx *c, const float *d_scores, int64_t n_rows, int64_t n, int64_t k, int32_t *d_idx_out) {
  for (int64_t r0 = 0; r0 < n_rows;) {
    const int64_t segs = std::min(sort_batch(n_rows, n), n_rows - r0);
    SortBufs b{};
    if (int rc = sort_layout(c, segs, n, 0, &b)) return rc;
    bh::ScoreKeyParams sp{};
    sp.scores = d_scores + r0 * n;
    sp.kv = b.A;
    sp.seg_stride = b.stride;
    sp.N = (int32_t)n;
    hipLaunchKernelGGL(bh::score_keys_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)segs), dim3(256), 0,
                       c->stream, sp);
    BH_HIP(hipGetLastError());
    if (int rc = radix_sort(c, b, b.A, b.B, segs, n)) return rc;
    hipLaunchKernelGGL(bh::take_front_kernel, dim3((unsigned)((k + 255) / 256), (unsigned)segs), dim3(256), 0,
                       c->stream, b.A, b.stride, (int32_t)k, d_idx_out + r0 * k);
    BH_HIP(hipGetLastError());
    r0 += segs;
  }
  return BH_OK;
}

int topk_impl(bh_ctx *c, const float *d_scores, int64_t n_rows, int64_t n, int64_t k, int32_t *d_idx_out) {
  if (n_rows < 0 || n < 0 || k < 0 || k > n) return fail(BH_ERR_BAD_ARG, "bh_topk: need 0 <= k <= n");
  if (n_rows == 0 || k == 0) return BH_OK;
  if (!d_scores || !d_idx_out) return fail(BH_ERR_BAD_ARG, "bh_topk: null device pointer");
  if (n > 0x3FFFFFFF || n_rows > 65535) return fail(BH_ERR_UNSUPPORTED, "bh_topk: shape too large");
  // more winners than one workgroup sorts in LDS, or many winners of a long row: the sort
  if (k > kTopkChunk || (n > kTopkChunk && k >= kTopkSortMinWinners)) return topk_by_sort(c, d_scores, n_rows, n, k, d_idx_out);

  if (!c->big_lds_set) {
    BH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(bh::topk_chunk_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, kTopkChunk * 8));
    c->big_lds_set = true;
  }
  // Long vectors.  One route: radix select of the k-th key, then a sort of the k winners in LDS -- one workgroup per
  // row, whatever its length.  The other: chunk sorts in levels (below), a row spread over many workgroups; worth it
  // where two levels are enough, i.e. few winners (PairScores' 200 of 40,000 pair scores).  A bitonic sort costs
  // P log^2 P, so those chunks are kept short: ~8 k candidates each, 2048 at least (4 x 40,000 -> 200: 119 us by
  // select, 158 us with 16384-candidate chunks, ~20 us with 4096-candidate ones)
  int64_t cs = 2048;
  while (cs < 8 * k && cs < kTopkChunk) cs <<= 1;
  if (n <= kTopkChunk) cs = kTopkChunk;  // a row one workgroup sorts whole
  const int64_t chunks0 = (n + cs - 1) / cs;
  // ... and rows of 4097 .. 16384 scores with at most half of them wanted: one workgroup's bitonic sort of the whole row
  // (padded to a power of two) costs P log^2 P whatever k is -- 68 us at 8192, 144 us at 16384 -- where four histogram
  // passes and a sort of the k winners take 42 and 52-67 us (round 5; at 4096 and below the sort is as fast)
  const bool select_mid = n > 4096 && n <= kTopkChunk && 2 * k <= n;
  if (select_mid || (n > kTopkChunk && chunks0 * k > kTopkChunk)) {
    bh::TopkSelectParams q{};
    q.scores = d_scores;
    q.out_idx = d_idx_out;
    q.n = (int32_t)n;
    q.k = (int32_t)k;
    q.P = 2;
    while (q.P < k) q.P <<= 1;
    const size_t lds = (size_t)q.P * 8;
    if (lds > 48 * 1024 && !c->select_lds_set) {
      BH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(bh::topk_select_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, kTopkChunk * 8));
      c->select_lds_set = true;
    }
    hipLaunchKernelGGL(bh::topk_select_kernel, dim3((unsigned)n_rows), dim3(1024), lds, c->stream, q);
    BH_HIP(hipGetLastError());
    return BH_OK;
  }
  // level 0 reads the caller's scores; every level keeps the best kk of each chunk of <= cs
  // candidates, until one chunk is left.  top-k of a union = top-k of the parts' top-k.
  const float *in_s = d_scores;
  const int32_t *in_i = nullptr;
  int64_t n_in = n;
  int flip = 0;
  for (;;) {
    const int64_t chunks = (n_in + cs - 1) / cs;
    const int64_t chunk = chunks == 1 ? n_in : cs;
    const int64_t kk = std::min<int64_t>(k, chunk);
    int64_t P = 2;
    while (P < chunk) P <<= 1;
    bh::TopkParams p{};
    p.scores = in_s;
    p.in_idx = in_i;
    p.n_in = (int32_t)n_in;
    p.chunk = (int32_t)chunk;
    p.P = (int32_t)P;
    p.kk = (int32_t)kk;
    if (chunks == 1) {
      p.out_idx = d_idx_out;
      p.out_scores = nullptr;
    } else {
      const int64_t need = n_rows * chunks * kk;
      if (need > c->cand_cap) {
        for (int i = 0; i < 2; ++i) {
          if (c->d_cand_s[i]) (void)hipFree(c->d_cand_s[i]);
          if (c->d_cand_i[i]) (void)hipFree(c->d_cand_i[i]);
          c->d_cand_s[i] = nullptr;
          c->d_cand_i[i] = nullptr;
        }
        c->cand_cap = 0;
        for (int i = 0; i < 2; ++i) {
          BH_HIP(hipMalloc(&c->d_cand_s[i], need * sizeof(float)));
          BH_HIP(hipMalloc(&c->d_cand_i[i], need * sizeof(int32_t)));
        }
        c->cand_cap = need;
      }
      p.out_idx = c->d_cand_i[flip];
      p.out_scores = c->d_cand_s[flip];
    }
    const int threads = (int)std::min<int64_t>(1024, std::max<int64_t>(64, P / 2));
    hipLaunchKernelGGL(bh::topk_chunk_kernel, dim3((unsigned)chunks, (unsigned)n_rows), dim3(threads),
                       (size_t)P * 8, c->stream, p);
    BH_HIP(hipGetLastError());
    if (chunks == 1) break;
    in_s = c->d_cand_s[flip];
    in_i = c->d_cand_i[flip];
    n_in = chunks * kk;
    flip ^= 1;
  }
  return BH_OK;
}


}  // namespace

namespace bhh {

// a deferred founder draw that found no launch to ride on is made now, in stream order
int flush_founders_draw(bh_ctx *c) {
  if (!c->founders_pending) return BH_OK;
  c->founders_pending = false;
  return launch_founders(c, c->fp_pending);
}

int flush_founders(bh_ctx *c) {
  if (int rc = flush_founders_draw(c)) return rc;
  return flush_gather(c);  // (its indices may be what that draw wrote)
}

}  // namespace bhh

extern "C" {

int bh_topk(bh_ctx *c, const float *d_scores, int64_t n_rows, int64_t n, int64_t k, int32_t *d_idx_out) {
  if (int rc = check_ctx(c)) return rc;
  return topk_impl(c, d_scores, n_rows, n, k, d_idx_out);
}

int bh_selection_actions(bh_ctx *c, const float *d_scores, int64_t n_envs, int64_t n, int64_t k, int64_t n_crosses,
                         uint32_t env_key[2], int32_t *d_actions) {
  if (int rc = check_ctx(c)) return rc;
  if (int rc = flush_founders(c)) return rc;
  if (!env_key || n_envs < 0 || n <= 0) return fail(BH_ERR_BAD_ARG, "bh_selection_actions: bad sizes");
  const int64_t n_pairs = k * (k - 1) / 2;
  if (k < 2 || k > n) return fail(BH_ERR_BAD_ARG, "bh_selection_actions: need 2 <= k <= n");
  if (n_crosses < 1 || n_crosses > n_pairs || n_crosses > n)
    return fail(BH_ERR_BAD_ARG, "bh_selection_actions: need 1 <= n_crosses <= min(k(k-1)/2, n)");
  if (n > 0x3FFFFFFF || n_pairs > 0x3FFFFFFF || n_envs > 65535)  // (element indices of the sort are 32-bit, tiles of 8192)
    return fail(BH_ERR_UNSUPPORTED, "bh_selection_actions: n and k(k-1)/2 < 2^30, n_envs <= 65535");
  const bool in_lds = n <= 8192 && n_pairs <= 8192;  // one workgroup per env keeps everything in LDS
  const int64_t e_total = c->envs_total > 0 ? c->envs_total : n_envs, e_off = c->envs_total > 0 ? c->env_offset : 0;
  if (e_off + n_envs > e_total) return fail(BH_ERR_BAD_ARG, "more envs than the env window holds");
  const bh::u32x2 nk = bh::split_at(env_key[0], env_key[1], (uint32_t)(e_total + 1), 0u, c->layout);
  if (n_envs > 0) {
    if (!d_scores || !d_actions) return fail(BH_ERR_BAD_ARG, "bh_selection_actions: null device pointer");
    if (in_lds) {
      bh::SelectionParams p{};
      p.scores = d_scores;
      p.actions = d_actions;
      p.key0 = env_key[0];
      p.key1 = env_key[1];
      p.E = (int32_t)n_envs;
      p.n = (int32_t)n;
      p.k = (int32_t)k;
      p.n_crosses = (int32_t)n_crosses;
      p.n_offspring = (int32_t)((n + n_crosses - 1) / n_crosses);
      p.n_pairs = (int32_t)n_pairs;
      p.P_top = 2;
      while (p.P_top < n) p.P_top <<= 1;
      p.P_pairs = 2;
      while (p.P_pairs < n_pairs) p.P_pairs <<= 1;
      p.rounds = (int32_t)std::ceil(3.0 * std::log((double)std::max<int64_t>(1, n_pairs)) / std::log(4294967295.0));
      p.layout = c->layout;
      p.E_total = (int32_t)e_total;
      p.env_offset = (int32_t)e_off;
      const int PK = std::max(p.P_top, p.P_pairs);
      const size_t lds = (size_t)PK * 8 + (size_t)p.P_pairs * 8 + (size_t)k * 4;
      if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(bh::selection_actions_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(BH_ERR_HIP, "hipFuncSetAttribute(selection LDS): %s", hipGetErrorString(e));
      }
      const int threads = (int)std::min<int64_t>(1024, std::max<int64_t>(64, PK / 2));
      hipLaunchKernelGGL(bh::selection_actions_kernel, dim3((unsigned)n_envs), dim3(threads), lds, c->stream, p);
      BH_HIP(hipGetLastError());
    } else {
      // the k best of every env (bh_topk's routes), then per batch of envs: the shuffle of the pair ranks as
      // radix sorts in HBM and the un-ranking of its first n_crosses
      void *best = c->d_best;
      const int rc0 = ensure_scratch(c, &best, &c->best_cap, n_envs * k * 4);
      c->d_best = best;
      if (rc0) return rc0;
      if (int rc = topk_impl(c, d_scores, n_envs, n, k, static_cast<int32_t *>(best))) return rc;
      for (int64_t e0 = 0; e0 < n_envs;) {
        const uint2 *perm = nullptr;
        int64_t stride = 0, segs = 0;
        if (int rc = shuffle_front(c, env_key, e_total, e_off + e0, n_envs - e0, n_pairs, n_crosses, &perm, &stride, &segs)) return rc;
        bh::PairActionParams q{};
        q.perm = perm;
        q.best = static_cast<const int32_t *>(best) + e0 * k;
        q.actions = d_actions + e0 * n * 2;
        q.seg_stride = stride;
        q.n = (int32_t)n;
        q.k = (int32_t)k;
        q.n_offspring = (int32_t)((n + n_crosses - 1) / n_crosses);
        hipLaunchKernelGGL(bh::pair_actions_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)segs), dim3(256), 0,
                           c->stream, q);
        BH_HIP(hipGetLastError());
        e0 += segs;
      }
    }
  }
  env_key[0] = nk.x;
  env_key[1] = nk.y;
  return BH_OK;
}

int bh_pair_actions(bh_ctx *c, const float *d_scores, int64_t n_envs, int64_t row_len, int64_t n, int32_t *d_actions) {
  if (int rc = check_ctx(c)) return rc;
  if (int rc = flush_founders(c)) return rc;
  if (n_envs < 0 || n <= 0 || row_len < n) return fail(BH_ERR_BAD_ARG, "bh_pair_actions: need n >= 1 scores per env and row_len >= n");
  if (n > 16384 || n_envs > 65535) return fail(BH_ERR_UNSUPPORTED, "bh_pair_actions: n <= 16384 and n_envs <= 65535");
  if (n_envs == 0) return BH_OK;
  if (!d_scores || !d_actions) return fail(BH_ERR_BAD_ARG, "bh_pair_actions: null device pointer");
  void *best = c->d_best;
  const int rc0 = ensure_scratch(c, &best, &c->best_cap, n_envs * n * 4);
  c->d_best = best;
  if (rc0) return rc0;
  if (int rc = topk_impl(c, d_scores, n_envs, row_len, n, static_cast<int32_t *>(best))) return rc;
  bh::PairScoreParams p{};
  p.scores = d_scores;
  p.best = static_cast<const int32_t *>(best);
  p.actions = d_actions;
  p.n = (int32_t)n;
  p.row_len = row_len;
  const size_t lds = (size_t)n * 8;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(bh::pair_score_actions_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return fail(BH_ERR_HIP, "hipFuncSetAttribute(pair scores LDS): %s", hipGetErrorString(e));
  }
  const int threads = (int)std::min<int64_t>(1024, std::max<int64_t>(64, (n + 63) / 64 * 64));
  hipLaunchKernelGGL(bh::pair_score_actions_kernel, dim3((unsigned)n_envs), dim3(threads), lds, c->stream, p);
  BH_HIP(hipGetLastError());
  return BH_OK;
}

int bh_winner_layout(bh_ctx *c, const int32_t *d_winners, int64_t k, int64_t n_local, int64_t world, int64_t rank,
                     int64_t cap, int32_t *d_send_idx, int32_t *d_pos, int32_t *d_counts, const int32_t *d_pairs,
                     int64_t n_pair_indices, int32_t *d_idx_out) {
  if (int rc = check_ctx(c)) return rc;
  if (k < 0 || n_local <= 0 || world <= 0 || rank < 0 || rank >= world || cap <= 0 || n_pair_indices < 0)
    return fail(BH_ERR_BAD_ARG, "bh_winner_layout: need k >= 0, n_local, world, cap >= 1 and 0 <= rank < world");
  if (world > bh::kMaxShardRanks) return fail(BH_ERR_UNSUPPORTED, "bh_winner_layout: at most %d ranks", bh::kMaxShardRanks);
  if (k > 0x3FFFFFFF || n_local * world > 0x7FFFFFFFll || cap * world > 0x7FFFFFFFll)
    return fail(BH_ERR_UNSUPPORTED, "bh_winner_layout: too many rows for 32-bit indexing");
  if (!d_send_idx || !d_pos || !d_counts || (k > 0 && !d_winners)) return fail(BH_ERR_BAD_ARG, "bh_winner_layout: null device pointer");
  if (n_pair_indices > 0 && (!d_pairs || !d_idx_out || k == 0)) return fail(BH_ERR_BAD_ARG, "bh_winner_layout: pairs need winners and an output");
  bh::WinnerLayoutParams p{d_winners, d_send_idx, d_pos, d_counts, (int32_t)k, (int32_t)n_local, (int32_t)world,
                           (int32_t)rank, (int32_t)cap};
  const int threads = (int)std::min<int64_t>(1024, std::max<int64_t>(64, round_up(k, 64)));
  const size_t lds = (size_t)world * (1 + threads / 64) * sizeof(int32_t);
  hipLaunchKernelGGL(bh::winner_layout_kernel, dim3(1), dim3(threads), lds, c->stream, p);
  BH_HIP(hipGetLastError());
  if (n_pair_indices > 0) {
    const unsigned grid = (unsigned)std::min<int64_t>(2048, (n_pair_indices + 255) / 256);
    hipLaunchKernelGGL(bh::remap_pairs_kernel, dim3(grid), dim3(256), 0, c->stream, d_pos, (int32_t)k, d_pairs, n_pair_indices,
                       d_idx_out);
    BH_HIP(hipGetLastError());
  }
  return BH_OK;
}

static int founders_device_impl(bh_ctx *c, uint32_t env_key[2], int64_t n_envs, int64_t n_germplasm, int64_t n,
                                int32_t *d_out, bool defer) {
  if (int rc = check_ctx(c)) return rc;
  if (int rc = flush_founders(c)) return rc;
  if (!env_key || n_envs < 0 || n < 0 || n > n_germplasm || (n_envs * n > 0 && !d_out))
    return fail(BH_ERR_BAD_ARG, "bh_founders_device: need 0 <= n <= n_germplasm");
  if (n_envs > 65535) return fail(BH_ERR_UNSUPPORTED, "bh_founders_device: too many envs");
  const int64_t e_total = c->envs_total > 0 ? c->envs_total : n_envs, e_off = c->envs_total > 0 ? c->env_offset : 0;
  if (e_off + n_envs > e_total) return fail(BH_ERR_BAD_ARG, "more envs than the env window holds");
  const bh::u32x2 nk = bh::split_at(env_key[0], env_key[1], (uint32_t)(e_total + 1), 0u, c->layout);
  if (n_envs * n > 0) {
    if (n_germplasm > 8192) {
      // beyond one workgroup's LDS: the shuffle rounds as radix sorts in HBM, envs side by side (sort_kernels.h)
      if (n_germplasm > 0x3FFFFFFF) return fail(BH_ERR_UNSUPPORTED, "bh_founders_device: germplasm of 2^30 rows and more");
      for (int64_t e0 = 0; e0 < n_envs;) {
        const uint2 *perm = nullptr;
        int64_t stride = 0, segs = 0;
        if (int rc = shuffle_front(c, env_key, e_total, e_off + e0, n_envs - e0, n_germplasm, n, &perm, &stride, &segs)) return rc;
        hipLaunchKernelGGL(bh::take_front_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)segs), dim3(256), 0,
                           c->stream, perm, stride, (int32_t)n, d_out + e0 * n);
        BH_HIP(hipGetLastError());
        e0 += segs;
      }
    } else {
      bh::FoundersParams p{};
      p.out = d_out;
      p.key0 = env_key[0];
      p.key1 = env_key[1];
      p.E = (int32_t)n_envs;
      p.N = (int32_t)n_germplasm;
      p.n = (int32_t)n;
      p.P = 2;
      while (p.P < n_germplasm) p.P <<= 1;
      p.rounds = (int32_t)std::ceil(3.0 * std::log((double)std::max<int64_t>(1, n_germplasm)) / std::log(4294967295.0));
      p.layout = c->layout;
      p.E_total = (int32_t)e_total;
      p.env_offset = (int32_t)e_off;
      if (int rc = (defer ? (c->fp_pending = p, c->founders_pending = true, BH_OK) : launch_founders(c, p))) return rc;
    }
  }
  env_key[0] = nk.x;
  env_key[1] = nk.y;
  return BH_OK;
}

int bh_founders_device(bh_ctx *c, uint32_t env_key[2], int64_t n_envs, int64_t n_germplasm, int64_t n,
                       int32_t *d_out) {
  return founders_device_impl(c, env_key, n_envs, n_germplasm, n, d_out, false);
}

int bh_founders_deferred(bh_ctx *c, uint32_t env_key[2], int64_t n_envs, int64_t n_germplasm, int64_t n,
                         int32_t *d_out) {
  return founders_device_impl(c, env_key, n_envs, n_germplasm, n, d_out, true);
}

}  // extern "C"

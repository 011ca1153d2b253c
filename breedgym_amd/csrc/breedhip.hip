// libbreedhip.so -- host side of the C ABI declared in include/breedhip.h: context, memory, settings and the host
// key plumbing.  HIP only: there is no CPU fallback in this library; without a gfx950 device every compute entry
// point fails with BH_ERR_NO_DEVICE / BH_ERR_HIP.  The other translation units are listed in ctx.h.
#include "ctx.h"

namespace bhh {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

int check_ctx(const bh_ctx *c) {
  if (!c) return fail(BH_ERR_BAD_ARG, "null context");
  hipError_t e = hipSetDevice(c->device);
  if (e != hipSuccess) return fail(BH_ERR_HIP, "hipSetDevice(%d): %s", c->device, hipGetErrorString(e));
  if (c->h_err) {
    const uint32_t w = *static_cast<volatile uint32_t *>(c->h_err);
    if (w == 0x53500001u)  // shuffle_sparse.h, kSpErrBucket
      return fail(BH_ERR_HIP, "an earlier draw (bh_founders_device / bh_selection_actions) met a bucket of more than 32768 "
                              "equal top halves among its Threefry words: its result, and every result since, is not to be trusted");
    if (w)
      return fail(BH_ERR_HIP, "an earlier launch reported a device-side error (code %u: a wait among the waves of a workgroup "
                              "ran out -- 1, 2: the mask role's hand-off; 4: the row waves' copy of the digit planes); its "
                              "results, and every result since, are not to be trusted", w);
  }
  return BH_OK;
}

int ensure_scratch(bh_ctx *c, void **buf, int64_t *cap, int64_t bytes) {
  if (*cap >= bytes) return BH_OK;
  BH_HIP(hipStreamSynchronize(c->stream));
  if (*buf) BH_HIP(hipFree(*buf));
  *buf = nullptr;
  *cap = 0;
  BH_HIP(hipMalloc(buf, (size_t)bytes));
  *cap = bytes;
  return BH_OK;
}

}  // namespace bhh

using namespace bhh;

extern "C" {

const char *bh_version(void) { return "breedhip 0.5.1 (gfx950)"; }
const char *bh_last_error(void) { return bhh::g_err; }

int bh_device_count(int *count) {
  if (!count) return fail(BH_ERR_BAD_ARG, "null count");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    return fail(BH_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
  }
  *count = n;
  return BH_OK;
}

int64_t bh_max_markers(void) {
  // 2^21: up to there a lane's int32 digit sums of the in-stream GEBV cannot wrap whatever the effects are (M / 64
  // markers per lane x dosage 2 x |digit| <= 2^15), and the tested shapes lie inside (the reference's largest:
  // 1,406,757 markers, scripts/time_cross.py:14).  Up to 2^18 markers a cross's site words stay in LDS; beyond, they
  // go through memory (mask_stream.h)
  return (int64_t)1 << 21;
}

int bh_create(int device, int64_t n_markers, int64_t n_traits, const float *recomb_vec,
              const float *effects, bh_ctx **out) {
  if (!out) return fail(BH_ERR_BAD_ARG, "null out");
  *out = nullptr;
  if (n_markers <= 0 || n_traits < 0 || !recomb_vec || (n_traits > 0 && !effects))
    return fail(BH_ERR_BAD_ARG, "bh_create: need n_markers > 0, n_traits >= 0 and non-null arrays");
  if (n_markers > bh_max_markers())
    return fail(BH_ERR_UNSUPPORTED, "bh_create: %lld markers exceed the supported %lld",
                (long long)n_markers, (long long)bh_max_markers());
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(BH_ERR_NO_DEVICE, "no HIP device visible (libbreedhip has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(BH_ERR_BAD_ARG, "device %d out of range [0,%d)", device, ndev);
  BH_HIP(hipSetDevice(device));

  bh_ctx *c = new (std::nothrow) bh_ctx();
  if (!c) return fail(BH_ERR_OOM, "host allocation failed");
  c->device = device;
  c->M = n_markers;
  c->T = n_traits;
  c->pitch = round_up(2 * n_markers, 16);
  c->half = (int32_t)((n_markers + 1) / 2);
  c->nwh = (c->half + 63) / 64;
  c->nw = (int32_t)((n_markers + 63) / 64);
  // Long rows start on 128-byte lines.  A 100,000-marker row is 25,008 bytes: every 128-byte piece the matrix-core GEBV
  // passes take of a row straddled two lines, every line was asked for twice -- the one-trait pass over 8192 rows takes
  // 40 us instead of 52 with rows on lines, configs[2]'s step 2-4 % less (profiles/r05_ab_pitch.txt; the mask rows keep
  // the same pitch for one stride in the row roles, their own alignment measures nothing).  Rows of up to 4 KB keep the
  // pitch their kernels were tuned at: the headline launch's stores do not care (profiles/r05_store_ceiling.txt).
  c->pw = c->nw > 256 ? (int32_t)round_up(c->nw, 8) : c->nw;
  c->zw = c->pw;
  c->nvec = (int32_t)(c->pitch / 16);
  c->effPitch = (int64_t)c->nvec * 8;

  // thresholds for the raw 32-bit compare: uniform(bits) < r  <=>  (bits >> 9) < t23  <=>
  // bits < (t23 << 9); t23 = 2^23 (r >= 1, always) does not fit and becomes an always-set bit.
  const int64_t nthr = (int64_t)(c->nwh + 16) * 64;  // + the trips the kernel prefetches ahead
  c->nthr = nthr;
  std::vector<uint32_t> thrA(nthr, 0u), thrB(nthr, 0u), ctrB(nthr, 0u);
  std::vector<uint64_t> alwA((size_t)c->nwh, 0ull), alwB((size_t)c->nwh, 0ull);
  bool any_always = false;
  for (int64_t j = 0; j < c->half; ++j) {
    const uint32_t ta = bh::threshold23(recomb_vec[j]);
    if (ta >> 23) {
      alwA[(size_t)(j >> 6)] |= 1ull << (j & 63);
      any_always = true;
    } else {
      thrA[j] = ta << 9;
    }
    if (j + c->half < n_markers) {
      ctrB[j] = (uint32_t)(j + c->half);
      const uint32_t tb = bh::threshold23(recomb_vec[j + c->half]);
      if (tb >> 23) {
        alwB[(size_t)(j >> 6)] |= 1ull << (j & 63);
        any_always = true;
      } else {
        thrB[j] = tb << 9;
      }
    }
  }
  std::vector<double> eff((size_t)std::max<int64_t>(1, n_traits * c->effPitch), 0.0);
  for (int64_t t = 0; t < n_traits; ++t)
    for (int64_t m = 0; m < n_markers; ++m) eff[t * c->effPitch + m] = (double)effects[m * n_traits + t];

  auto cleanup = [&](int rc) {
    bh_destroy(c);
    return rc;
  };
#define BH_HIP_C(call)                                                                          \
  do {                                                                                          \
    hipError_t e_ = (call);                                                                     \
    if (e_ != hipSuccess)                                                                       \
      return cleanup(fail(e_ == hipErrorOutOfMemory ? BH_ERR_OOM : BH_ERR_HIP, "%s failed: %s", \
                          #call, hipGetErrorString(e_)));                                       \
  } while (0)
  BH_HIP_C(hipHostMalloc(reinterpret_cast<void **>(&c->h_err), sizeof(uint32_t), hipHostMallocMapped));
  *c->h_err = 0u;
  BH_HIP_C(hipHostGetDevicePointer(reinterpret_cast<void **>(&c->d_err), c->h_err, 0));
  BH_HIP_C(hipMalloc(&c->d_spare, 16384));
  BH_HIP_C(hipMalloc(&c->d_thrA, nthr * sizeof(uint32_t)));
  BH_HIP_C(hipMalloc(&c->d_thrB, nthr * sizeof(uint32_t)));
  BH_HIP_C(hipMalloc(&c->d_eff, eff.size() * sizeof(double)));
  BH_HIP_C(hipMemcpy(c->d_thrA, thrA.data(), nthr * sizeof(uint32_t), hipMemcpyHostToDevice));
  BH_HIP_C(hipMemcpy(c->d_thrB, thrB.data(), nthr * sizeof(uint32_t), hipMemcpyHostToDevice));
  BH_HIP_C(hipMalloc(&c->d_ctrB, nthr * sizeof(uint32_t)));
  BH_HIP_C(hipMemcpy(c->d_ctrB, ctrB.data(), nthr * sizeof(uint32_t), hipMemcpyHostToDevice));
  if (any_always) {
    BH_HIP_C(hipMalloc(&c->d_alwA, alwA.size() * sizeof(uint64_t)));
    BH_HIP_C(hipMalloc(&c->d_alwB, alwB.size() * sizeof(uint64_t)));
    BH_HIP_C(hipMemcpy(c->d_alwA, alwA.data(), alwA.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
    BH_HIP_C(hipMemcpy(c->d_alwB, alwB.data(), alwB.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
  }
  BH_HIP_C(hipMemcpy(c->d_eff, eff.data(), eff.size() * sizeof(double), hipMemcpyHostToDevice));
  if (n_traits > 0) {
    std::vector<uint8_t> pos8((size_t)c->nvec, 0);
    for (int64_t m = 0; m < n_markers; ++m)
      if (effects[m * n_traits] > 0.0f) pos8[(size_t)(m >> 3)] |= (uint8_t)(1u << (m & 7));
    BH_HIP_C(hipMalloc(&c->d_pos8, pos8.size()));
    BH_HIP_C(hipMemcpy(c->d_pos8, pos8.data(), pos8.size(), hipMemcpyHostToDevice));
  }
  {
    std::vector<double> effsum((size_t)c->effPitch, 0.0);
    for (int64_t m = 0; m < n_markers; ++m)
      for (int64_t t = 0; t < n_traits; ++t) effsum[(size_t)m] += (double)effects[m * n_traits + t];
    BH_HIP_C(hipMalloc(&c->d_effsum, effsum.size() * sizeof(double)));
    BH_HIP_C(hipMemcpy(c->d_effsum, effsum.data(), effsum.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  // ---- fixed-point effects for the MFMA GEBV (gebv_mfma.h): 6 balanced base-256 digits of
  //      round(eff * 2^s_t), s_t the largest shift that keeps |.| <= 2^46
  if (n_traits > 0) {
    constexpr int kTraitsPerPack = kGebvCols / kGebvDigits;  // 10
    c->n_chunks = (int32_t)((n_markers + 63) / 64);
    std::vector<double> scale((size_t)n_traits);
    std::vector<int> shift((size_t)n_traits);
    for (int64_t t = 0; t < n_traits; ++t) {
      double emax = 0.0, esum = 0.0;
      for (int64_t m = 0; m < n_markers; ++m) {
        const double a = std::fabs((double)effects[m * n_traits + t]);
        emax = std::max(emax, a);
        esum += a;
      }
      int sft = 0;
      if (emax > 0.0 && std::isfinite(esum)) {
        int e2, s2;
        std::frexp(emax, &e2);        // emax = f * 2^e2, f in [0.5, 1)
        std::frexp(2.0 * esum, &s2);  // the largest |GEBV| any genotype can reach: dosage <= 2
        // every fixed-point effect fits the digits (|.| < 2^46), and the 56-bit sum field of the cells in which
        // the digit sums meet (gebv_bits.h, GebvOut: the top byte counts arrivals) holds any row:
        // sum_m 2 |round(e_m 2^sft)| <= 2 esum 2^sft + M < 2^54 + 2^18 < 2^55
        sft = std::min(46 - e2, 54 - s2);
      }
      shift[(size_t)t] = sft;
      scale[(size_t)t] = std::ldexp(1.0, -sft);
    }
    BH_HIP_C(hipMalloc(&c->d_scale, (size_t)n_traits * sizeof(double)));
    BH_HIP_C(hipMemcpy(c->d_scale, scale.data(), (size_t)n_traits * sizeof(double), hipMemcpyHostToDevice));
    {  // the same fixed-point effects as three balanced base-65536 digits, 8 markers (one vector) per 16 bytes; every
      // plane ends with a zero block (index nvec): where the lanes that redo a row's last vector point their loads
      const int64_t nvd = (int64_t)c->nvec + 1;
      std::vector<int16_t> dig((size_t)n_traits * nvd * 3 * 8, (int16_t)0);
      for (int64_t t = 0; t < n_traits; ++t)
        for (int64_t m = 0; m < n_markers; ++m) {
          long long fix = std::llround(std::ldexp((double)effects[m * n_traits + t], shift[(size_t)t]));
          for (int j = 0; j < 3; ++j) {
            long long d = ((fix % 65536) + 65536) % 65536;
            if (d >= 32768) d -= 65536;
            fix = (fix - d) / 65536;
            dig[(size_t)(((t * 3 + j) * nvd + (m >> 3)) * 8 + (m & 7))] = (int16_t)d;
          }
        }
      BH_HIP_C(hipMalloc(&c->d_dig16, dig.size() * sizeof(int16_t)));
      BH_HIP_C(hipMemcpy(c->d_dig16, dig.data(), dig.size() * sizeof(int16_t), hipMemcpyHostToDevice));
    }
    const int64_t packs = (n_traits + kTraitsPerPack - 1) / kTraitsPerPack;
    std::vector<int8_t> pack((size_t)c->n_chunks * 4096);
    for (int64_t g = 0; g < packs; ++g) {
      std::fill(pack.begin(), pack.end(), (int8_t)0);
      const int64_t t0 = g * kTraitsPerPack, tn = std::min<int64_t>(kTraitsPerPack, n_traits - t0);
      for (int64_t m = 0; m < n_markers; ++m) {
        const int64_t ch = m >> 6, w = m & 63, h = w >> 5, q = (w >> 4) & 1, j = w & 15;
        for (int64_t t = 0; t < tn; ++t) {
          long long fix = std::llround(std::ldexp((double)effects[m * n_traits + t0 + t], shift[(size_t)(t0 + t)]));
          for (int d = 0; d < kGebvDigits; ++d) {
            long long dig = ((fix % 256) + 256) % 256;
            if (dig >= 128) dig -= 256;
            fix = (fix - dig) / 256;
            const int64_t col = t * kGebvDigits + d, nt = col >> 5, r = col & 31, lane = r + 32 * h;
            pack[(size_t)((((ch * 2 + nt) * 2 + q) * 64 + lane) * 16 + j)] = (int8_t)dig;
          }
        }
      }
      void *dB = nullptr;
      BH_HIP_C(hipMalloc(&dB, pack.size()));
      c->d_Bpack.push_back(dB);
      BH_HIP_C(hipMemcpy(dB, pack.data(), pack.size(), hipMemcpyHostToDevice));
    }
    // the same digits in gebv_bits_kernel's order: [mc][q 4][ntile 2][j 4][lane (col, h)][16 markers]
    // <-> marker 512 mc + 256 h + 64 q + 16 j + i (operand byte 4 (i & 3) + (i >> 2)); a macro-chunk of zeros behind the last one
    // (the staging runs two steps ahead)
    c->n_mc = (int32_t)((n_markers + 511) / 512);
    std::vector<int8_t> bits((size_t)(c->n_mc + 1) * 32768);
    for (int64_t g = 0; g < packs; ++g) {
      std::fill(bits.begin(), bits.end(), (int8_t)0);
      const int64_t t0 = g * kTraitsPerPack, tn = std::min<int64_t>(kTraitsPerPack, n_traits - t0);
      for (int64_t m = 0; m < n_markers; ++m) {
        const int64_t mc = m >> 9, rem = m & 511, h = rem >> 8, q = (rem >> 6) & 3, j = (rem >> 4) & 3, i = rem & 15;
        for (int64_t t = 0; t < tn; ++t) {
          long long fix = std::llround(std::ldexp((double)effects[m * n_traits + t0 + t], shift[(size_t)(t0 + t)]));
          for (int d = 0; d < kGebvDigits; ++d) {
            long long dig = ((fix % 256) + 256) % 256;
            if (dig >= 128) dig -= 256;
            fix = (fix - dig) / 256;
            const int64_t col = t * kGebvDigits + d, nt = col >> 5, lane = (col & 31) + 32 * h;
            // operand byte of local marker i: dword i & 3, byte i >> 2 (dosage16_from_bits)
            bits[(size_t)((((((mc * 4 + q) * 2 + nt) * 4 + j) * 64 + lane) * 16) + 4 * (i & 3) + (i >> 2))] = (int8_t)dig;
          }
        }
      }
      void *dB = nullptr;
      BH_HIP_C(hipMalloc(&dB, bits.size()));
      c->d_Bbits.push_back(dB);
      BH_HIP_C(hipMemcpy(dB, bits.data(), bits.size(), hipMemcpyHostToDevice));
    }
  }
#undef BH_HIP_C
  *out = c;
  return BH_OK;
}

int bh_destroy(bh_ctx *c) {
  if (!c) return BH_OK;
  (void)hipSetDevice(c->device);
  for (int i = 0; i < 2; ++i)
    if (c->d_Z[i]) (void)hipFree(c->d_Z[i]);
  if (c->h_err) (void)hipHostFree(c->h_err);
  if (c->d_S) (void)hipFree(c->d_S);
  if (c->d_spare) (void)hipFree(c->d_spare);
  if (c->d_sort) (void)hipFree(c->d_sort);
  if (c->d_best) (void)hipFree(c->d_best);
  if (c->d_F) (void)hipFree(c->d_F);
  if (c->d_thrA) (void)hipFree(c->d_thrA);
  if (c->d_thrB) (void)hipFree(c->d_thrB);
  if (c->d_ctrB) (void)hipFree(c->d_ctrB);
  if (c->d_alwA) (void)hipFree(c->d_alwA);
  for (auto *b : c->d_Bpack)
    if (b) (void)hipFree(b);
  for (auto *b : c->d_Bbits)
    if (b) (void)hipFree(b);
  if (c->d_scale) (void)hipFree(c->d_scale);
  if (c->d_dig16) (void)hipFree(c->d_dig16);
  if (c->d_partial) (void)hipFree(c->d_partial);
  if (c->d_gsum) (void)hipFree(c->d_gsum);
  if (c->d_pool_gebv) (void)hipFree(c->d_pool_gebv);
  if (c->d_pos8) (void)hipFree(c->d_pos8);
  if (c->d_ohp) (void)hipFree(c->d_ohp);
  if (c->d_alwB) (void)hipFree(c->d_alwB);
  if (c->d_eff) (void)hipFree(c->d_eff);
  if (c->d_effsum) (void)hipFree(c->d_effsum);
  for (int i = 0; i < 2; ++i) {
    if (c->d_cand_s[i]) (void)hipFree(c->d_cand_s[i]);
    if (c->d_cand_i[i]) (void)hipFree(c->d_cand_i[i]);
  }
  delete c;
  return BH_OK;
}

int bh_set_stream(bh_ctx *c, void *hip_stream) {
  if (int rc = check_ctx(c)) return rc;
  hipStream_t s = reinterpret_cast<hipStream_t>(hip_stream);
  if (s == c->stream) return BH_OK;
  // what the old stream still owes -- masks made ahead for the next call, a deferred founder draw, scratch in
  // use -- is finished before the new stream may touch it
  if (int rc = flush_founders(c)) return rc;
  BH_HIP(hipStreamSynchronize(c->stream));
  c->stream = s;
  c->primed_reward = nullptr;  // a buffer primed on the old stream is primed again
  return BH_OK;
}

int bh_sync(bh_ctx *c) {
  if (int rc = check_ctx(c)) return rc;
  if (int rc = flush_founders(c)) return rc;
  BH_HIP(hipStreamSynchronize(c->stream));
  return check_ctx(c);  // ... and whatever the launches that just finished had to report
}

int bh_clear_error(bh_ctx *c, uint32_t *was) {
  if (!c) return fail(BH_ERR_BAD_ARG, "null context");
  BH_HIP(hipSetDevice(c->device));
  const uint32_t w = c->h_err ? *static_cast<volatile uint32_t *>(c->h_err) : 0u;
  if (was) *was = w;
  // what was deferred is made now (its results are as untrusted as everything since the event, but nothing is left
  // pending), then the stream is drained: no launch is in flight while the invariants are restored
  if (int rc = flush_founders(c)) return rc;
  BH_HIP(hipStreamSynchronize(c->stream));
  if (c->d_gsum && c->gsum_cap > 0) BH_HIP(hipMemsetAsync(c->d_gsum, 0, (size_t)c->gsum_cap, c->stream));
  if (c->d_S && c->s_cap > 0) BH_HIP(hipMemsetAsync(c->d_S, 0, (size_t)c->s_cap, c->stream));
  c->pf.valid = false;          // masks made ahead by a launch that may have broken
  c->hint.valid = false;
  c->primed_reward = nullptr;   // a reward block primed by it is primed again
  c->primed_envs = 0;
  BH_HIP(hipStreamSynchronize(c->stream));
  if (c->h_err) *static_cast<volatile uint32_t *>(c->h_err) = 0u;
  return BH_OK;
}

int bh_debug_inject_error(bh_ctx *c, uint32_t code, int dirty) {
  if (!c) return fail(BH_ERR_BAD_ARG, "null context");
  BH_HIP(hipSetDevice(c->device));
  BH_HIP(hipStreamSynchronize(c->stream));
  if (dirty) {
    if (!c->d_gsum || c->gsum_cap < 8) return fail(BH_ERR_BAD_ARG, "bh_debug_inject_error: no GEBV pass has run yet (no cells to dirty)");
    // every 64th cell of the final set: an arrival count and a sum nobody will complete
    const unsigned long long junk = (3ull << 56) | 0x123456789ull;
    for (int64_t off = 0; off + 8 <= c->gsum_cap && off < (int64_t)64 * 8 * 64; off += 64 * 8)
      BH_HIP(hipMemcpy(static_cast<char *>(c->d_gsum) + off, &junk, sizeof junk, hipMemcpyHostToDevice));
  }
  if (c->h_err) *static_cast<volatile uint32_t *>(c->h_err) = code;
  return BH_OK;
}

int bh_debug_count_dirty_cells(bh_ctx *c, int64_t *count) {
  if (!c || !count) return fail(BH_ERR_BAD_ARG, "bh_debug_count_dirty_cells: null argument");
  BH_HIP(hipSetDevice(c->device));
  BH_HIP(hipStreamSynchronize(c->stream));
  *count = 0;
  if (!c->d_gsum || c->gsum_cap <= 0) return BH_OK;
  std::vector<unsigned long long> cells((size_t)(c->gsum_cap / 8));
  BH_HIP(hipMemcpy(cells.data(), c->d_gsum, cells.size() * 8, hipMemcpyDeviceToHost));
  for (unsigned long long v : cells) *count += v != 0ull;
  return BH_OK;
}

int bh_release_scratch(bh_ctx *c) {
  if (int rc = check_ctx(c)) return rc;
  if (int rc = flush_founders(c)) return rc;
  BH_HIP(hipStreamSynchronize(c->stream));
  auto drop = [](void *&p, int64_t &cap) {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  };
  drop(c->d_sort, c->sort_cap);
  drop(c->d_best, c->best_cap);
  drop(c->d_gsum, c->gsum_cap);  // (zero whenever no launch is in flight: nothing is lost)
  drop(c->d_pool_gebv, c->pool_gebv_cap);
  drop(c->d_ohp, c->ohp_cap);
  {
    void *p = c->d_partial;
    drop(p, c->partial_cap);
    c->d_partial = nullptr;
  }
  for (int i = 0; i < 2; ++i) {
    if (c->d_cand_s[i]) (void)hipFree(c->d_cand_s[i]);
    if (c->d_cand_i[i]) (void)hipFree(c->d_cand_i[i]);
    c->d_cand_s[i] = nullptr;
    c->d_cand_i[i] = nullptr;
  }
  c->cand_cap = 0;
  return BH_OK;
}

int64_t bh_row_pitch(const bh_ctx *c) { return c ? c->pitch : 0; }
int64_t bh_n_markers(const bh_ctx *c) { return c ? c->M : 0; }
int64_t bh_n_traits(const bh_ctx *c) { return c ? c->T : 0; }

int bh_alloc(bh_ctx *c, int64_t bytes, void **dptr) {
  if (int rc = check_ctx(c)) return rc;
  if (!dptr || bytes < 0) return fail(BH_ERR_BAD_ARG, "bh_alloc: bad arguments");
  *dptr = nullptr;
  BH_HIP(hipMalloc(dptr, (size_t)std::max<int64_t>(bytes, 16)));
  return BH_OK;
}

int bh_free(bh_ctx *c, void *dptr) {
  if (int rc = check_ctx(c)) return rc;
  if (dptr) BH_HIP(hipFree(dptr));
  return BH_OK;
}

int bh_memcpy_h2d(bh_ctx *c, void *dst, const void *src, int64_t bytes) {
  if (int rc = check_ctx(c)) return rc;
  if (bytes < 0 || (bytes > 0 && (!dst || !src))) return fail(BH_ERR_BAD_ARG, "bh_memcpy_h2d: bad arguments");
  if (bytes) BH_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyHostToDevice, c->stream));
  return BH_OK;
}

int bh_memcpy_d2h(bh_ctx *c, void *dst, const void *src, int64_t bytes) {
  if (int rc = check_ctx(c)) return rc;
  if (int rc = flush_founders(c)) return rc;
  if (bytes < 0 || (bytes > 0 && (!dst || !src))) return fail(BH_ERR_BAD_ARG, "bh_memcpy_d2h: bad arguments");
  if (bytes) BH_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToHost, c->stream));
  BH_HIP(hipStreamSynchronize(c->stream));
  return BH_OK;
}

int bh_upload_population(bh_ctx *c, const uint8_t *host, int64_t rows, uint8_t *d_dst, int validate) {
  if (int rc = check_ctx(c)) return rc;
  if (rows < 0 || (rows > 0 && (!host || !d_dst))) return fail(BH_ERR_BAD_ARG, "bh_upload_population: bad arguments");
  if (rows == 0) return BH_OK;
  const int64_t w = 2 * c->M;
  if (validate) {
    const int64_t n = rows * w;
    uint8_t bad = 0;
    for (int64_t i = 0; i < n; ++i) bad |= host[i];
    if (bad & 0xFE) return fail(BH_ERR_BAD_GENOTYPE, "genotype bytes must be 0 or 1");
  }
  if (c->pitch != w) BH_HIP(hipMemsetAsync(d_dst, 0, (size_t)(rows * c->pitch), c->stream));
  BH_HIP(hipMemcpy2DAsync(d_dst, (size_t)c->pitch, host, (size_t)w, (size_t)w, (size_t)rows,
                          hipMemcpyHostToDevice, c->stream));
  BH_HIP(hipStreamSynchronize(c->stream));  // the host buffer may be pageable and short-lived
  return BH_OK;
}

int bh_download_population(bh_ctx *c, const uint8_t *d_src, int64_t rows, uint8_t *host) {
  if (int rc = check_ctx(c)) return rc;
  if (int rc = flush_founders(c)) return rc;
  if (rows < 0 || (rows > 0 && (!host || !d_src))) return fail(BH_ERR_BAD_ARG, "bh_download_population: bad arguments");
  if (rows == 0) return BH_OK;
  const int64_t w = 2 * c->M;
  BH_HIP(hipMemcpy2DAsync(host, (size_t)w, d_src, (size_t)c->pitch, (size_t)w, (size_t)rows,
                          hipMemcpyDeviceToHost, c->stream));
  BH_HIP(hipStreamSynchronize(c->stream));
  return BH_OK;
}

void bh_key_from_seed(uint64_t seed, uint32_t key[2]) {
  key[0] = 0u;  // jax with x64 disabled drops the high word
  key[1] = (uint32_t)(seed & 0xFFFFFFFFull);
}

static bool layout_ok(int layout) { return layout == bh::LAYOUT_ORIGINAL || layout == bh::LAYOUT_PARTITIONABLE; }

int bh_set_env_window(bh_ctx *c, int64_t env_offset, int64_t envs_total) {
  if (!c) return fail(BH_ERR_BAD_ARG, "null context");
  if (env_offset < 0 || envs_total < 0 || (envs_total > 0 && env_offset >= envs_total) || envs_total > 0x3FFFFFFF)
    return fail(BH_ERR_BAD_ARG, "bh_set_env_window: need 0 <= env_offset < envs_total (or 0, 0)");
  c->env_offset = envs_total > 0 ? env_offset : 0;
  c->envs_total = envs_total;
  // masks made ahead stay: they carry the key window they were drawn for, and a call under another window does
  // not match them (cross_impl)
  return BH_OK;
}

int bh_set_cross_window(bh_ctx *c, int64_t cross_offset, int64_t crosses_total) {
  if (!c) return fail(BH_ERR_BAD_ARG, "null context");
  if (cross_offset < 0 || crosses_total < 0 || (crosses_total > 0 && cross_offset >= crosses_total) ||
      crosses_total > 0x3FFFFFFF)
    return fail(BH_ERR_BAD_ARG, "bh_set_cross_window: need 0 <= cross_offset < crosses_total (or 0, 0)");
  c->cross_offset = crosses_total > 0 ? cross_offset : 0;
  c->crosses_total = crosses_total;
  return BH_OK;
}

int bh_set_threefry_layout(bh_ctx *c, int layout) {
  if (!c) return fail(BH_ERR_BAD_ARG, "null context");
  if (!layout_ok(layout)) return fail(BH_ERR_BAD_ARG, "bh_set_threefry_layout: unknown layout %d", layout);
  if (layout != c->layout) c->pf.valid = false;  // masks made ahead belong to the other bit stream
  c->layout = layout;
  return BH_OK;
}

int bh_set_mutation(bh_ctx *c, float probability) {
  if (!c) return fail(BH_ERR_BAD_ARG, "null context");
  if (!(probability >= 0.0f && probability <= 1.0f))
    return fail(BH_ERR_BAD_ARG, "bh_set_mutation: probability must be in [0, 1]");
  c->mutation = probability;
  c->mut_thr = bh::threshold23(probability);
  return BH_OK;
}

int bh_key_split(const uint32_t key[2], int64_t num, int layout, uint32_t *out) {
  if (!key || !out || num < 0 || num > 0x7FFFFFFF || !layout_ok(layout))
    return fail(BH_ERR_BAD_ARG, "bh_key_split: bad arguments");
  for (int64_t k = 0; k < num; ++k) {
    const bh::u32x2 s = bh::split_at(key[0], key[1], (uint32_t)num, (uint32_t)k, layout);
    out[2 * k] = s.x;
    out[2 * k + 1] = s.y;
  }
  return BH_OK;
}

int bh_permutation(const uint32_t key_in[2], int64_t n, int layout, int32_t *out) {
  if (!key_in || n < 0 || (n > 0 && !out) || n > 0x7FFFFFFF || !layout_ok(layout))
    return fail(BH_ERR_BAD_ARG, "bh_permutation: bad arguments");
  // jax `_shuffle`: ceil(3 ln n / ln(2^32 - 1)) rounds of a stable sort by fresh random words.
  // Sorting (random word << 32 | current position) with an ordinary sort IS that stable sort.
  uint32_t key[2] = {key_in[0], key_in[1]};
  const int rounds = (int)std::ceil(3.0 * std::log((double)std::max<int64_t>(1, n)) / std::log(4294967295.0));
  thread_local std::vector<uint64_t> kv;
  thread_local std::vector<int32_t> x, y;
  kv.resize((size_t)n);
  x.resize((size_t)n);
  y.resize((size_t)n);
  std::iota(x.begin(), x.end(), 0);
  const int64_t L = n + (n & 1), h = L / 2;
  for (int rd = 0; rd < rounds; ++rd) {
    const bh::u32x2 nk = bh::split_at(key[0], key[1], 2u, 0u, layout), sub = bh::split_at(key[0], key[1], 2u, 1u, layout);
    key[0] = nk.x;
    key[1] = nk.y;
    if (layout == bh::LAYOUT_PARTITIONABLE) {
      for (int64_t i = 0; i < n; ++i)
        kv[(size_t)i] = ((uint64_t)bh::bits_word_partitionable(sub.x, sub.y, (uint32_t)i) << 32) | (uint64_t)i;
    } else {
      for (int64_t i = 0; i < h; ++i) {  // random_bits(sub, n): odd n is padded with a 0 counter
        const uint32_t c1 = (i + h < n) ? (uint32_t)(i + h) : 0u;
        const bh::u32x2 o = bh::threefry2x32(sub.x, sub.y, (uint32_t)i, c1);
        kv[(size_t)i] = ((uint64_t)o.x << 32) | (uint64_t)i;
        if (i + h < n) kv[(size_t)(i + h)] = ((uint64_t)o.y << 32) | (uint64_t)(i + h);
      }
    }
    std::sort(kv.begin(), kv.end());
    for (int64_t i = 0; i < n; ++i) y[(size_t)i] = x[(size_t)(uint32_t)kv[(size_t)i]];
    x.swap(y);
  }
  std::copy(x.begin(), x.end(), out);
  return BH_OK;
}

int bh_founders(uint32_t env_key[2], int64_t n_envs, int64_t n_germplasm, int64_t n, int layout, int32_t *out) {
  if (!env_key || n_envs < 0 || n < 0 || n > n_germplasm || (n_envs * n > 0 && !out))
    return fail(BH_ERR_BAD_ARG, "bh_founders: need 0 <= n <= n_germplasm");
  std::vector<uint32_t> keys((size_t)(2 * (n_envs + 1)));
  if (int rc = bh_key_split(env_key, n_envs + 1, layout, keys.data())) return rc;
  std::vector<int32_t> perm((size_t)n_germplasm);
  for (int64_t e = 0; e < n_envs; ++e) {
    if (int rc = bh_permutation(&keys[(size_t)(2 * (e + 1))], n_germplasm, layout, perm.data())) return rc;
    std::copy(perm.begin(), perm.begin() + n, out + e * n);
  }
  env_key[0] = keys[0];
  env_key[1] = keys[1];
  return BH_OK;
}

}  // extern "C"

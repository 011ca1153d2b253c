// The row role of the long-row launch (packed_long_kernel) with its loads taken off the critical path.
//
// What bounded the register-staged stream (packed_stream_role<0>) on long rows was not bytes: with the loads removed
// its stores alone run at 6.1 TB/s (8192 x 100k: 300 us), with them the launch's row role took 560 us -- also when
// every parent row came out of the L2 (profiles/r02_exp_long_row_loads.txt).  A CU's vector memory pipeline is in
// order: a wave's next words queue behind every store the CU's waves have issued, the wave cannot issue its next stores
// before those words are back, and gfx950's single vmcnt counts loads and stores alike -- so each wave had one trip of
// stores in flight and paid a full queue-plus-memory latency per trip, whatever the number of waves.
//
// Here the words of a row come by LDS-DMA (global_load_lds_dwordx4: no VGPR destination) in 1 KB chunks of each of
// the three arrays (parent 0, parent 1, mask: 512 words = 4096 markers), kDmaSlots chunks ahead, into a ring in LDS
// that belongs to the wave alone (no barrier, the mask waves next door never wait).  A trip waits for ITS chunk with a
// counted vmcnt that leaves the later chunks and the stores of the last trips in flight, reads its words back
// (ds_read_u16: lane l takes word 64 u + l, the same mapping as before, so every store stays contiguous), re-arms the
// slot for the chunk kDmaSlots ahead and issues its stores.  Stores in flight per wave: kDmaSlots - 1 trips of 9 KB
// instead of one of 4.5 (vmcnt holds 63 operations: a trip is nine stores and three DMAs).
//
// The compiler is kept out of the counting: the LDS reads are inline asm (hipcc would otherwise wait for the newest
// DMA, i.e. for everything), the DMA block is pinned between scheduling barriers, and the count assumes the least
// number of later operations any schedule inside a trip could leave.  That only holds if every trip issues the same
// ST stores and DT DMAs: none of them is predicated (lanes past a row's end store to a spare line; lane 0 of a DMA is
// always inside the row), which tests/test_isa.py checks in the emitted ISA of the shipped kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bh {

constexpr int kDmaSlots = 6;  // chunks in flight per row wave
constexpr int kDmaSlotBytes = 3 * 1024;                   // parent 0 | parent 1 | mask, 512 words each (a gather: 1 KB)
constexpr int kDmaRingBytes = kDmaSlots * kDmaSlotBytes;  // per row wave of a cross
constexpr int kDmaGatherRingBytes = kDmaSlots * 1024;     // per row wave of a gather

// s_waitcnt vmcnt(n) alone (gfx9 encoding: vmcnt in bits 3:0 and 15:14, expcnt 6:4 and lgkmcnt 11:8 left at "any")
__host__ __device__ constexpr int vmcnt_only(int n) { return (n & 0xF) | ((n >> 4) << 14) | 0x0F70; }

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *glb_ptr_t;

// eight words of one array of a slot: lane l gets words 64 u + l, u = 0..7 (`addr` = slot base + 2 l)
template <int OFF>
__device__ __forceinline__ void dma_read8(uint32_t addr, uint32_t (&w)[8]) {
  asm volatile(
      "ds_read_u16 %0, %8 offset:%9\n\t"
      "ds_read_u16 %1, %8 offset:%10\n\t"
      "ds_read_u16 %2, %8 offset:%11\n\t"
      "ds_read_u16 %3, %8 offset:%12\n\t"
      "ds_read_u16 %4, %8 offset:%13\n\t"
      "ds_read_u16 %5, %8 offset:%14\n\t"
      "ds_read_u16 %6, %8 offset:%15\n\t"
      "ds_read_u16 %7, %8 offset:%16\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(w[4]), "=&v"(w[5]), "=&v"(w[6]), "=&v"(w[7])
      : "v"(addr), "n"(OFF), "n"(OFF + 128), "n"(OFF + 256), "n"(OFF + 384), "n"(OFF + 512), "n"(OFF + 640), "n"(OFF + 768),
        "n"(OFF + 896)
      : "memory");
}

// the same slot the other way round: lane l gets the 16 bytes (words 8 l .. 8 l + 7) of each array
__device__ __forceinline__ void dma_read128(uint32_t addr, uint4 &a, uint4 &b, uint4 &z) {
  asm volatile(
      "ds_read_b128 %0, %3\n\t"
      "ds_read_b128 %1, %3 offset:1024\n\t"
      "ds_read_b128 %2, %3 offset:2048\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(a), "=&v"(b), "=&v"(z)
      : "v"(addr)
      : "memory");
}

__device__ __forceinline__ void dma_read128(uint32_t addr, uint4 &a) {
  asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(a) : "v"(addr) : "memory");
}

// The byte-layout observation of long rows is written once and not read again by the launches that follow (1.6 GB
// at 8192 x 100k, far beyond the 256 MB Infinity Cache): as a non-temporal stream it leaves the cache to the bit-plane
// offspring, which the GEBV pass and the next launch's parent reads want (configs[2]: launch 0.628 -> 0.592 ms, GEBV
// pass 90 -> 81 us).
__device__ __forceinline__ void store_obs(char *dst, const uint4 v) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 x = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(x, reinterpret_cast<u32x4 *>(dst));
}

// cross16 on two packed words at once (no bit crosses a marker, let alone a word)
__device__ __forceinline__ uint32_t cross32(uint32_t p0, uint32_t p1, uint32_t z) {
  return bfi(0x55555555u, bfi(z, p0 >> 1, p0), bfi(z, p1, p1 << 1));
}

// `ring`: kDmaRingBytes (cross) / kDmaGatherRingBytes (gather: one array, the row itself) of LDS owned by this wave.
// The wave's p.seq rows are ONE stream of chunks: the DMA cursor runs kDmaSlots chunks ahead of the store cursor across
// row boundaries too (a fresh prologue per row would have to wait for every store of the row before); parent indices
// come by scalar loads, which are not counted by vmcnt.  No GEBV in the stream.
template <int MODE, bool OBS>
__device__ __forceinline__ void packed_dma_role(const PackedParams &p, const int block, const int wave, char *ring) {
  static_assert(MODE == MODE_CROSS || MODE == MODE_GATHER, "packed rows: cross or gather");
  constexpr bool CROSS = MODE == MODE_CROSS;
  constexpr int SLOT_BYTES = CROSS ? kDmaSlotBytes : 1024;
  constexpr int S = kDmaSlots;
  // vmcnt holds 63 operations per wave: the fewer a trip needs, the more bytes a wave keeps in flight.  The packed
  // offspring words of a chunk go out as ONE 16-byte store per lane (lane l: words 8 l .. 8 l + 7, crossed two words
  // per instruction), the observation as eight (lane l: word 64 u + l, a contiguous KB per instruction)
  constexpr int ST = OBS ? 9 : 1;
  constexpr int DT = CROSS ? 3 : 1;  // DMA instructions per chunk
  static_assert(S >= 2 && S <= 6 && (S - 1) * (ST + DT) <= 63, "the counted wait must fit vmcnt's six bits");
  typedef const int32_t __attribute__((address_space(4))) *scalar_i32_ptr;
  const int lane = threadIdx.x & 63;
  const int seq = p.seq > 1 ? p.seq : 1;
  __builtin_amdgcn_s_setprio(kLongRowPrio);
  const int first = __builtin_amdgcn_readfirstlane((xcd_row(block, p.n_blocks) * 4 + wave) * seq);
  if (first >= p.n_rows) return;
  const int rows_here = p.n_rows - first < seq ? p.n_rows - first : seq;
  const uint32_t ring_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)ring;
  const int npieces = p.nvp >> 3;            // 16-byte pieces per packed row
  const int n_chunks = (npieces + 63) >> 6;  // 1 KB chunks per row and array
  const int total = rows_here * n_chunks;
  const scalar_i32_ptr sidx = (scalar_i32_ptr)(uintptr_t)p.idx;

  // ---- DMA cursor: (row dr, chunk dk) and the three source rows of row dr
  int dr = 0, dk = 0;
  const char *a0 = nullptr, *a1 = nullptr, *az = nullptr;
  auto dma_row = [&](int c) {
    if constexpr (CROSS) {
      const int e = c / p.n_out, i = c - e * p.n_out;
      const int i0 = norm_index(sidx[2 * (int64_t)c], p.n_pool), i1 = norm_index(sidx[2 * (int64_t)c + 1], p.n_pool);
      a0 = reinterpret_cast<const char *>(p.src + ((int64_t)e * p.n_pool + i0) * p.nvp);
      a1 = reinterpret_cast<const char *>(p.src + ((int64_t)e * p.n_pool + i1) * p.nvp);
      az = reinterpret_cast<const char *>(p.Z + (int64_t)(p.rng_shared ? i : c) * p.zpitch);
    } else {
      const int i0 = p.idx ? norm_index(sidx[c], p.n_pool) : c;
      a0 = reinterpret_cast<const char *>(p.src + (int64_t)i0 * p.nvp);
    }
  };
  dma_row(first);
  // the cursor's chunk -> slot, then advance; past the last chunk of the last row that chunk again, so that every
  // trip issues the same number of operations (the counted waits rely on it)
  auto dma_next = [&](int slot) {
    const int piece = 64 * dk + lane;
    char *dst = ring + slot * SLOT_BYTES;  // wave-uniform
    // never predicated (the counted waits rely on every trip issuing DT of these): lanes past the row's end fetch
    // its last piece again, into LDS words of their own that only the spare-line stores ever look at
    const uint32_t off = 16u * (uint32_t)(piece < npieces ? piece : npieces - 1);
    __builtin_amdgcn_global_load_lds((glb_ptr_t)(a0 + off), (lds_ptr_t)dst, 16, 0, 0);
    if constexpr (CROSS) {
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(a1 + off), (lds_ptr_t)(dst + 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(az + off), (lds_ptr_t)(dst + 2048), 16, 0, 0);
    }
    if (dk + 1 < n_chunks) {
      ++dk;
    } else if (dr + 1 < rows_here) {
      ++dr;
      dk = 0;
      dma_row(first + dr);
    }
  };
  for (int k = 0; k < S; ++k) dma_next(k);
  __builtin_amdgcn_sched_barrier(0);

  // ---- store cursor
  int sr = 0, t = 0, slot = 0;
  char *dp = reinterpret_cast<char *>(p.outp + (int64_t)first * p.nvp);
  char *db = OBS ? reinterpret_cast<char *>(p.outb + (int64_t)first * p.pitch) : nullptr;
  for (int j = 0; j < total; ++j) {
    // wait until at most N later operations are pending.  Behind chunk j < S - 1: the S-1-j chunks of the prologue
    // and j whole trips (DMA + stores each); from then on chunk j was asked for in trip j - S (or the prologue)
    // and S - 1 whole trips have been issued since (that trip's own stores are not counted on: a stricter wait
    // than the schedule as written needs, whatever order the compiler gives a trip's operations)
#define BH_DMA_WAIT(T_) __builtin_amdgcn_s_waitcnt(vmcnt_only(DT * (S - 1 - (T_)) + (T_) * (ST + DT)))
    if (j >= S - 1)
      BH_DMA_WAIT(S - 1);
    else if (j == 0)
      BH_DMA_WAIT(0);
    else if (j == 1)
      BH_DMA_WAIT(S > 1 ? 1 : 0);
    else if (j == 2)
      BH_DMA_WAIT(S > 2 ? 2 : 0);
    else if (j == 3)
      BH_DMA_WAIT(S > 3 ? 3 : 0);
    else
      BH_DMA_WAIT(S > 4 ? 4 : 0);
#undef BH_DMA_WAIT
    const uint32_t slot_lds = ring_lds + (uint32_t)slot * SLOT_BYTES;
    uint4 pa, pb, pz;
    if constexpr (CROSS)
      dma_read128(slot_lds + 16u * (uint32_t)lane, pa, pb, pz);
    else
      dma_read128(slot_lds + 16u * (uint32_t)lane, pa);
    uint32_t wa[8], wb[8], wz[8];
    if constexpr (OBS) {
      const uint32_t rd = slot_lds + 2u * (uint32_t)lane;
      dma_read8<0>(rd, wa);
      if constexpr (CROSS) {
        dma_read8<1024>(rd, wb);
        dma_read8<2048>(rd, wz);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    dma_next(slot);
    __builtin_amdgcn_sched_barrier(0);
    // parents hold zero bits for the markers past M and zero words up to the pitch: so does what is crossed from them
    uint4 po = pa;
    if constexpr (CROSS)
      po = make_uint4(cross32(pa.x, pb.x, pz.x), cross32(pa.y, pb.y, pz.y), cross32(pa.z, pb.z, pz.z), cross32(pa.w, pb.w, pz.w));
    auto word = [&](int u) -> uint32_t {
      if constexpr (CROSS) return cross16(wa[u], wb[u], wz[u]);
      return wa[u];
    };
    const int piece = 64 * t + lane;
    if (512 * t + 512 <= p.nvec) {  // a whole chunk of real words: no lane tests
      *reinterpret_cast<uint4 *>(dp + 16u * (uint32_t)piece) = po;
      if constexpr (OBS) {
        const uint32_t v16 = 16u * (uint32_t)(512 * t + lane);
#pragma unroll
        for (int u = 0; u < 8; ++u) store_obs(db + (v16 + 1024u * u), expand_vector(word(u)));
      }
    } else {
      // A row's last chunk.  The counted waits above rely on EVERY trip issuing ST stores: a store whose lanes are
      // all past the row's end would be branched over (s_cbranch_execz) and never counted, so no store is
      // predicated here -- the lanes past the end send theirs to a spare line instead (p.spare, 16 KB, never read).
      char *spare = p.spare + 16u * (uint32_t)lane;
      *reinterpret_cast<uint4 *>(piece < npieces ? dp + 16u * (uint32_t)piece : spare + 8192) = po;
      if constexpr (OBS) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int w = 512 * t + 64 * u + lane;
          store_obs(w < p.nvec ? db + 16u * (uint32_t)w : spare + 1024 * u, expand_vector(word(u)));
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    slot = slot + 1 == S ? 0 : slot + 1;
    if (++t == n_chunks) {  // next row
      t = 0;
      ++sr;
      dp = reinterpret_cast<char *>(p.outp + (int64_t)(first + sr) * p.nvp);
      if constexpr (OBS) db = reinterpret_cast<char *>(p.outb + (int64_t)(first + sr) * p.pitch);
    }
  }
}

}  // namespace bh

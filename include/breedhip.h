/*
 * breedhip.h -- C ABI of libbreedhip.so, the MI355X (gfx950) implementation of the one
 * data-parallel hot path of younik/breedgym: vectorised reset()/step() =
 * parent gather -> meiosis/cross -> GEBV -> reward / top-k.
 *
 * The reference has no FFI of its own; its seam is the Python object
 * `self.simulator` (a chromax.Simulator) called from
 *     breedgym/vector/vec_env.py:45-50,77,114,133     (VecBreedGym)
 *     breedgym/breedgym.py:36-38,117,143,233,240      (BreedGym)
 * Every entry point below names the reference call it replaces.  Plain C types
 * only: pointers, sizes, status codes.  Device pointers are raw HIP device
 * addresses (e.g. torch.Tensor.data_ptr()); nothing here knows about torch.
 *
 * Conventions
 *   - every function returns BH_OK (0) or a negative bh_status; the message of
 *     the last failure on the calling thread is at bh_last_error().
 *   - genotypes are 1 byte per allele in {0,1}, laid out (rows, M, 2) like the
 *     reference's bool arrays, with a row pitch of bh_row_pitch() bytes
 *     (= 2*M rounded up to 16; equal to 2*M, i.e. plain contiguous, when
 *     M % 8 == 0).  Pad bytes are zero.
 *   - all work is enqueued on the context's HIP stream (default: the null
 *     stream, which is also torch's default stream) and is asynchronous;
 *     bh_sync() or any stream sync makes results visible to the host.
 *   - a context is bound to one device and is not re-entrant; use one per
 *     thread / per process (one process per GPU is the intended deployment).
 */
#ifndef BREEDHIP_H
#define BREEDHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum bh_status {
  BH_OK = 0,
  BH_ERR_BAD_ARG = -1,     /* null pointer, negative size, bad mode            */
  BH_ERR_OOM = -2,         /* hipMalloc / host allocation failed               */
  BH_ERR_HIP = -3,         /* any other HIP runtime error                      */
  BH_ERR_UNSUPPORTED = -4, /* shape outside what the kernels handle            */
  BH_ERR_NO_DEVICE = -5,   /* no gfx950 device visible                         */
  BH_ERR_BAD_GENOTYPE = -6 /* a genotype byte other than 0/1                   */
} bh_status;

/* RNG stream assignment inside one bh_cross call (see DESIGN.md "RNG contract"). */
typedef enum bh_rng_mode {
  BH_RNG_SHARED_ENVS = 0, /* every env gets the same gamete keys: what jax.vmap over the
                             un-batched simulator key does in vec_env.py:75-77       */
  BH_RNG_INDEPENDENT = 1, /* the E*n crosses form one flat chromax cross() batch      */
  BH_RNG_DOUBLE_HAPLOID = 2 /* flag, OR-ed in: each row is ONE gamete of its parent written to
                               both allele slots (`Simulator.double_haploid`,
                               breeding_programs_env.py:39-41); keys = split(key, B + 1),
                               row b uses keys[1 + b]; pass the same index twice in d_idx */
} bh_rng_mode;

/* Counter layout of JAX's threefry PRNG (jax/_src/prng.py).  ORIGINAL: jax_threefry_partitionable = False,
 * the default up to jax 0.4.x and the one the reference's seeded expectations were recorded under
 * (tests/test_vec.py:132 ...): split / random_bits hash iota() cut in two halves, one call per two words.
 * PARTITIONABLE: the default since jax 0.5: element i comes from its own call with counter (0, i); a new
 * key is both output words, a 32-bit draw their XOR.  Same primitive, different bit streams. */
typedef enum bh_threefry_layout { BH_THREEFRY_ORIGINAL = 0, BH_THREEFRY_PARTITIONABLE = 1 } bh_threefry_layout;

typedef struct bh_ctx bh_ctx; /* opaque */

const char *bh_version(void);
const char *bh_last_error(void);
int bh_device_count(int *count);

/* ---- context = the simulator's device-side constants -------------------------------
 * Replaces `Simulator(genetic_map=..., trait_names=...)` + `.device`
 * (vec_env.py:45-46, breedgym.py:36,40).  recomb_vec: (M) f32 host, the per-marker
 * recombination probability r[m] (first marker of a chromosome already 0.5);
 * effects: (M, T) f32 host, row-major, marker effects of the T selected traits. */
int bh_create(int device, int64_t n_markers, int64_t n_traits, const float *recomb_vec,
              const float *effects, bh_ctx **out);
int bh_destroy(bh_ctx *ctx);
/* Move the context to another HIP stream (hipStream_t; NULL = the null stream).  Work already enqueued on the old
 * stream is waited for first.  Every device buffer handed to later calls must be safe to use on that stream: with a
 * caching allocator (torch) allocate it while that stream is current, or record it to the stream
 * (Tensor.record_stream) -- a block the allocator recycles on another stream while a launch still reads it is the
 * caller's race, not the library's. */
int bh_set_stream(bh_ctx *ctx, void *hip_stream);
int bh_sync(bh_ctx *ctx);
/* Device-side errors are STICKY.  The kernels report what the host cannot see at launch time through one 32-bit word
 * in host memory that every entry point that takes a context reads first; once it is non-zero every call fails with
 * BH_ERR_HIP ("an earlier launch reported a device-side error ...") until bh_clear_error().  What sets it:
 *   bit 0 (1)   mask role, four waves per cross taking crosses in turns: the wait for the previous cross's prefix pass
 *               ran out (csrc/row_kernels.h, mask_role, SYNC_COUNTER, `done`)
 *   bit 1 (2)   mask role: the prefix wave's wait for the other three waves' site words ran out (`counter`)
 *   bit 2 (4)   row waves of a two-role workgroup: the wait for the others' share of the LDS copy of the digit planes
 *               ran out (csrc/packed_kernels.h, packed_stream_role, DIGLDS)
 *   0x53500001  a draw met a bucket of more than 32768 equal top halves among its Threefry words
 *               (csrc/shuffle_sparse.h; bh_founders_device / bh_selection_actions)
 * The three waits are bounded spins (2^22 ... 2^24 sleeps) among the waves of ONE workgroup, all of which are resident
 * by construction, so they end; should one run out all the same, the wave sets its bit and CONTINUES on data that may
 * be incomplete -- the launch ends (no hang), and everything it and later launches wrote is not to be trusted: masks
 * made ahead, GEBV, rewards.  No such event has been observed; the word exists so that it could not pass unseen.
 *
 * bh_clear_error: recover the context instead of destroying it.  Waits for the stream, runs what was deferred
 * (founder draws, a gather asked for ahead), then restores every invariant a broken launch may have left violated --
 * the cells of the matrix-core GEBV passes and the site words of the long-row mask role are zeroed again ("all zero
 * whenever no launch is in flight"), masks made ahead and a primed reward block are dropped (the next call makes its
 * own), a pending bh_hint_next_cross is forgotten -- and clears the word.  *was (may be null) receives the word that was
 * set (0: nothing to clear; the scrub is done all the same).  Results of the calls between the event and the clear
 * stay untrusted: redo them.  The reference has no counterpart (a JAX error is an exception; there is no device-side
 * state to repair). */
int bh_clear_error(bh_ctx *ctx, uint32_t *was);
/* TEST HOOK, not part of the path: sets the device-side error word to `code` and, if `dirty` != 0, leaves a non-zero
 * value in a cell of the matrix-core GEBV passes -- the state a launch that died half-way would leave -- so that the
 * recovery above can be tested (tests/test_gpu_parity.py).  Bypasses the sticky check. */
int bh_debug_inject_error(bh_ctx *ctx, uint32_t code, int dirty);
/* TEST HOOK: waits for the stream and counts the cells of the matrix-core GEBV passes that are not zero.  The passes
 * keep no counters beside their cells and never clear them up front: "all zero whenever no launch is in flight" is the
 * invariant every launch relies on (whoever completes a cell carries its sum on and leaves it zero), so after any
 * completed call the count must be 0 -- for every trait count and ragged row count (tests/test_gpu_parity.py). */
int bh_debug_count_dirty_cells(bh_ctx *ctx, int64_t *count);
/* Frees the scratch buffers a context has grown for its larger calls (the draws' and top-k's sort scratch, candidate
 * lists, the GEBV passes' cells and slabs); they are allocated again on demand.  Waits for the stream first.  The
 * reference has nothing to replace here (JAX's allocator owns its temporaries); a long-lived process that made one
 * 65,536-individual selection and goes back to small envs gets its ~0.5-1 GB back with this. */
int bh_release_scratch(bh_ctx *ctx);
int64_t bh_row_pitch(const bh_ctx *ctx);
int64_t bh_n_markers(const bh_ctx *ctx);
int64_t bh_n_traits(const bh_ctx *ctx);
/* largest marker count a context takes (2^21; the reference's largest shape, scripts/time_cross.py:14, has
 * 1,406,757 markers) */
int64_t bh_max_markers(void);

/* ---- device memory for callers without a HIP runtime of their own ------------------ */
int bh_alloc(bh_ctx *ctx, int64_t bytes, void **dptr);
int bh_free(bh_ctx *ctx, void *dptr);
int bh_memcpy_h2d(bh_ctx *ctx, void *dst, const void *src, int64_t bytes);
int bh_memcpy_d2h(bh_ctx *ctx, void *dst, const void *src, int64_t bytes); /* synchronises */

/* `simulator.load_population` / `jax.device_put(germplasm)` (vec_env.py:48-52):
 * host (rows, M, 2) contiguous bytes -> device rows at bh_row_pitch().  With
 * `validate` != 0 every byte must be 0 or 1 (BH_ERR_BAD_GENOTYPE otherwise). */
int bh_upload_population(bh_ctx *ctx, const uint8_t *host, int64_t rows, uint8_t *d_dst,
                         int validate);
/* device rows -> host (rows, M, 2) contiguous; synchronises. */
int bh_download_population(bh_ctx *ctx, const uint8_t *d_src, int64_t rows, uint8_t *host);

/* ---- the hot path -------------------------------------------------------------------
 * bh_cross = `populations[arange, actions]` + `vmap(simulator.cross)` + `GEBV_model`
 * (vec_env.py:88-91,132-134; single env: breedgym.py:142-143,233) fused in one kernel:
 * the parent gather is never materialised.
 *   d_pool  (n_envs, pool_rows, pitch) u8   the current populations
 *   d_idx   (n_envs, n_out, 2) int32        env-local parent rows; negative wraps once,
 *                                           out of range is clamped (JAX gather rules)
 *   d_out   (n_envs, n_out, pitch) u8       offspring; out[e,i,m,p] comes from parent p
 *   d_gebv  (n_envs, n_out, T) f32 or NULL  GEBV of the offspring
 *   d_reward (n_envs) f32 or NULL           max over an env's (n_out, T) GEBV block -- the reward of
 *                                           vec_env.py:97.  Where the GEBV is summed inside the launch
 *                                           (up to 8 traits, short rows) the row waves note their values
 *                                           with atomic maxima into this buffer, which must hold -inf
 *                                           first: it does when it was the d_reward_next of the call
 *                                           before (and nothing wrote to it since), otherwise the
 *                                           library fills it first; elsewhere a reduction kernel follows
 *   d_reward_next (n_envs) f32 or NULL      left holding -inf by this launch, for the caller to pass as
 *                                           d_reward of its next call (saves that fill).  The library recognises
 *                                           it by address, length and pitch: between the two calls the caller must
 *                                           neither write to it nor free it (a freed block can come back from an
 *                                           allocator at the same address with other contents)
 *   reward_pitch                            floats from one env's word to the next in d_reward and
 *                                           d_reward_next (>= 1; 64 gives every env a cache line of its own,
 *                                           so that the atomics of different envs do not queue behind each other)
 *   sim_key host, in/out                    the simulator's threefry key; advanced once
 *                                           per call like `Simulator.cross` does
 * d_out must not alias d_pool. */
int bh_cross(bh_ctx *ctx, const uint8_t *d_pool, int64_t n_envs, int64_t pool_rows,
             const int32_t *d_idx, int64_t n_out, uint32_t sim_key[2], int rng_mode,
             uint8_t *d_out, float *d_gebv, float *d_reward, float *d_reward_next, int64_t reward_pitch);

/* Run-ahead of the crossover masks.  The masks of a generation depend on the key chain and
 * the batch size only, never on the parents, so every bh_cross launch also makes the masks of
 * the NEXT generation: its workgroups take two roles that share each CU, a memory-bound one
 * that streams the current generation and a VALU-bound one that runs Threefry for the next.
 * A call whose key or batch size was not predicted (first call, re-seed, new shape) makes its
 * masks in stream order first.  Results never depend on this switch.  Default: enabled. */
int bh_set_mask_prefetch(bh_ctx *ctx, int enable);

/* Row gather + GEBV: `_random_selection` / `germplasm[selected_indices]` + `get_info`
 * (vec_env.py:22-27,126-129; breedgym.py:121-125).  d_idx (n_rows) int32 into d_src
 * (src_rows, pitch); d_out (n_rows, pitch); d_gebv (n_rows, T) or NULL. */
int bh_gather(bh_ctx *ctx, const uint8_t *d_src, int64_t src_rows, const int32_t *d_idx,
              int64_t n_rows, uint8_t *d_out, float *d_gebv);

/* `simulator.GEBV_model(population)` (vec_env.py:133): d_pop (rows, pitch) -> (rows, T) f32.
 * GEBV[i,t] = sum_m (pop[i,m,0] + pop[i,m,1]) * effects[m,t], accumulated in f64. */
int bh_gebv(bh_ctx *ctx, const uint8_t *d_pop, int64_t rows, float *d_gebv);

/* `np.max(infos["GEBV"], axis=(1, 2))` (vec_env.py:97): (n_envs, per_env) f32 -> (n_envs) f32. */
int bh_reward_max(bh_ctx *ctx, const float *d_gebv, int64_t n_envs, int64_t per_env,
                  float *d_reward);

/* `jax.lax.top_k(scores, k)[1]` (vec_wrappers.py:64; Simulator.select): for each of n_rows
 * score vectors of length n, the indices of the k largest in descending order, ties to the
 * lower index (order = XLA's total order of the bit patterns: -0 below +0).  d_scores (n_rows, n) f32;
 * d_idx_out (n_rows, k) int32.  Any k <= n (above 16384 winners, or more than 512 of a row longer than 16384:
 * a radix sort of the row in HBM). */
int bh_topk(bh_ctx *ctx, const float *d_scores, int64_t n_rows, int64_t n, int64_t k,
            int32_t *d_idx_out);

/* `SelectionScores._convert_actions` (vec_wrappers.py:61-79) for all envs in one launch: per env the
 * k best by score, a random subset of n_crosses of their k(k-1)/2 diallel pairs (without
 * replacement, key = split(env_key, E+1)[1+e]), each repeated ceil(n / n_crosses) times and cut to
 * n.  d_scores (E, n) f32 -> d_actions (E, n, 2) int32; env_key is advanced like the wrapper's
 * `random_key` (keys[0]).  Up to n <= 8192 and k(k-1)/2 <= 8192 one workgroup per env does it all in LDS; beyond
 * (the reference's default k = n // 10 at 2048 individuals already is: 20,706 pairs; 65,536 individuals give
 * 21.5 M pairs and three shuffle rounds) only the n_crosses wanted positions are followed backwards through the
 * shuffle rounds (bh_founders_device has the routes).  Asynchronous like every launch.  k(k-1)/2 < 2^30. */
int bh_selection_actions(bh_ctx *ctx, const float *d_scores, int64_t n_envs, int64_t n, int64_t k,
                         int64_t n_crosses, uint32_t env_key[2], int32_t *d_actions);

/* `PairScores._convert_actions` (vec_wrappers.py:100-112) for all envs: the n best of every env's pair scores
 * (jax.lax.top_k on `action.flatten()`: row_len scores per env -- n * n for the documented (E, n, n) action, but the
 * reference flattens whatever it is handed, e.g. the (n, T, E, n, T) block its own tests/test_vec.py:198 feeds with T
 * traits), offspring per pair = ceil(softmax(their scores) * n), pairs repeated that often and cut at n.
 * d_scores (E, row_len) f32 -> d_actions (E, n, 2) int32, pair = (flat index // n, flat index % n) -- beyond n - 1 for
 * row_len > n * n, which the cross entry points clamp as JAX's gather does.
 * The softmax is taken in f32 with its sum in f64, rounded once (XLA's f32 reduction order is not reproduced: a sum
 * that differs in its last bit can move ceil() across an integer for a score exactly on the boundary).
 * n <= 16384, n <= row_len < 2^30. */
int bh_pair_actions(bh_ctx *ctx, const float *d_scores, int64_t n_envs, int64_t row_len, int64_t n, int32_t *d_actions);

/* Population sharding (the reference has no call to replace here -- its multi-device mode is one worker
 * process per device with whole observations piped to a parent, vec_env.py:150-236 -- this is that role for ONE
 * population cut row-wise over `world` ranks, rank r owning rows [r n_local, (r + 1) n_local)).  d_winners (k) int32:
 * the global top-k's row indices, best first, identical on every rank.  The winners' rows travel by one all-gather of
 * `cap`-row blocks; this call derives, on the device, from the list alone:
 *   d_counts (world + 1)  winners each rank owns; [world] = the largest count.  A count above `cap` means the blocks
 *                         were too small: nothing of this call may be used, call again with cap >= d_counts[world]
 *   d_send_idx (cap)      local rows this rank puts into its block, in list order (padding: row 0 again)
 *   d_pos (k)             row of winner i in the gathered (world * cap)-row buffer: owner * cap + earlier winners of
 *                         the same owner
 *   d_idx_out             unless n_pair_indices == 0: d_pos[d_pairs[i]] for the n_pair_indices parent indices of this
 *                         rank's crosses (indices into the winners' list; JAX's retrieval rule on k) -- what bh_cross*
 *                         then reads the gathered buffer with.  Rows are never permuted. */
int bh_winner_layout(bh_ctx *ctx, const int32_t *d_winners, int64_t k, int64_t n_local, int64_t world, int64_t rank,
                     int64_t cap, int32_t *d_send_idx, int32_t *d_pos, int32_t *d_counts, const int32_t *d_pairs,
                     int64_t n_pair_indices, int32_t *d_idx_out);

/* ---- bit-plane storage behind the byte-layout API ------------
 * The observation format of the reference stays what it is -- (rows, M, 2) one byte per allele
 * (vec_env.py:57-62) -- but a population a VecBreedGym keeps between steps can live packed, 1 bit
 * per allele: a packed row is uint16 P[bh_packed_pitch() / 2]; bit i of P[v] is byte i of the 16-byte
 * vector v of the byte-layout row (markers 8v .. 8v+7: even bits allele 0, odd bits allele 1; i.e.
 * numpy packbits(row_bytes, bitorder="little")), zero beyond M.  bh_packed_pitch() = 16 bytes per 64 markers, rounded
 * up to whole 128-byte lines when that is more than 4 KB (M > 16384: long rows start on cache lines -- a 100,000-marker
 * row is 25,088 bytes apart from the next, not 25,008); the padding is zero in every row the library writes.  Ask for
 * the pitch, do not compute it.
 * A step then reads M/2 bytes of parents per cross instead of 4M, and still hands out the same
 * byte-layout offspring when asked to. */
int64_t bh_packed_pitch(const bh_ctx *ctx);
/* byte-layout rows (n_rows, pitch) -> packed rows (n_rows, packed_pitch); bit 0 of each byte counts. */
int bh_pack(bh_ctx *ctx, const uint8_t *d_rows, int64_t n_rows, uint16_t *d_packed);
/* bh_cross on a packed pool (vec_env.py:88-91,132-134 like bh_cross).  d_out_packed (n_envs, n_out,
 * packed_pitch) receives the offspring; d_out_bytes, unless NULL, receives the same offspring in
 * the byte layout (n_envs, n_out, pitch) -- the observation `step` returns.  Same key chain, same
 * masks, same offspring as bh_cross on the unpacked pool. */
int bh_cross_packed(bh_ctx *ctx, const uint16_t *d_pool, int64_t n_envs, int64_t pool_rows,
                    const int32_t *d_idx, int64_t n_out, uint32_t sim_key[2], int rng_mode,
                    uint16_t *d_out_packed, uint8_t *d_out_bytes, float *d_gebv, float *d_reward,
                    float *d_reward_next, int64_t reward_pitch);
/* The masks a launch makes ahead are those of a call of the same shape -- right for a step loop.  A scheme that
 * alternates shapes (breeding_programs_env.py:47-56: 200 crosses, then 20,000 doubled haploids, then 200 crosses ...)
 * says before a call what the call AFTER it will be, and that call finds its masks made: n_envs, n_out and rng_mode as
 * they will be passed.  One hint serves one call; without one the next call is assumed to repeat this one; a wrong
 * hint costs time, never results.  Ignored while a sharding window is set. */
int bh_hint_next_cross(bh_ctx *ctx, int64_t n_envs, int64_t n_out, int rng_mode);

/* `VecBreedGym.step` (vec_env.py:79-107) on output blocks bound once: what a step loop repeats -- four output
 * pointers, sizes, pitches -- is handed over once, and a step is bh_cross_packed with the few arguments that change.
 * The caller owns the blocks and decides which one a step writes (it alone knows which of them somebody still
 * looks at): `depth` generation slots, each
 *   d_state[s]   (n_envs * n_out, packed_pitch)  the generation's bit planes
 *   d_obs[s]     (n_envs * n_out, pitch)         its byte-layout observation  (d_obs may be NULL: never wanted)
 *   d_gebv[s]    (n_envs, n_out, T) f32
 * and, apart from them (a step primes the block of the NEXT step's reward, whose slot is not known yet),
 *   d_reward[r]  (n_envs) f32 at reward_pitch, r < n_rewards.
 * depth 0 unbinds.  The tables are copied; the blocks must outlive the binding. */
int bh_step_bind(bh_ctx *ctx, int64_t n_envs, int64_t n_out, int64_t depth, void *const *d_state, void *const *d_obs,
                 void *const *d_gebv, int64_t n_rewards, void *const *d_reward, int64_t reward_pitch);
/* One step into generation slot `slot`: parents d_pool (n_envs, pool_rows, packed_pitch) -- normally the state of the
 * slot written last --, d_actions (n_envs, n_out, 2).  flags: bh_rng_mode in the low two bits, BH_STEP_OBSERVATION
 * (write the slot's byte-layout observation).  reward_slot >= 0: max GEBV per env into that reward block
 * (vec_env.py:97); prime_slot >= 0: that reward block is left holding -inf for the step that will use it
 * (bh_cross_packed's d_reward_next); -1: none.  Everything else as bh_cross_packed. */
enum { BH_STEP_OBSERVATION = 4 };
int bh_step(bh_ctx *ctx, int64_t slot, int64_t reward_slot, int64_t prime_slot, const uint16_t *d_pool,
            int64_t pool_rows, const int32_t *d_actions, uint32_t sim_key[2], int flags);
/* bh_gather on packed rows (vec_env.py:126-129).  d_idx NULL = rows 0..n_rows-1 in order, which
 * makes this the unpack (d_out_bytes only) and the GEBV of packed rows (d_gebv only).  Any of
 * d_out_packed, d_out_bytes, d_gebv may be NULL. */
int bh_gather_packed(bh_ctx *ctx, const uint16_t *d_src, int64_t src_rows, const int32_t *d_idx,
                     int64_t n_rows, uint16_t *d_out_packed, uint8_t *d_out_bytes, float *d_gebv);
/* bh_gather_packed, asked for AHEAD of a bh_cross_packed / bh_step call: the rows are written by the next two-role
 * launch of the context (its grid grows by the gather's workgroups; bit-plane rows out, GEBV of one or two traits in
 * the stream) or, failing that, by a launch of their own before anything else of this context runs.  The reference
 * has no counterpart (its reset, vec_env.py:109-129, follows its step); VecBreedGym's horizon step uses it when the
 * founders of the reset behind it are already drawn: that step's crosses write no observation (the step hands out
 * the next episode's first one, vec_env.py:102-107), the gather's rows do, and one launch does both instead of two
 * half-empty ones.  What the gather writes must not be anything that call reads or writes. */
int bh_gather_packed_deferred(bh_ctx *ctx, const uint16_t *d_src, int64_t src_rows, const int32_t *d_idx,
                              int64_t n_rows, uint16_t *d_out_packed, uint8_t *d_out_bytes, float *d_gebv);

/* Selection indices of the reference (breedgym/utils/index_functions.py:21-78): the optimal haploid of
 * every individual and its value.  d_pop (n_rows, pitch).  n_blocks == 0: marker m of the optimal
 * haploid is any(alleles) where effects[m, 0] > 0 and all(alleles) elsewhere (:66-78); otherwise
 * d_block_start (n_blocks + 1) int32 marker offsets on the device and, per block, the haplotype with the
 * larger sum_m sum_t allele * effects[m, t] (the first on ties) supplies the block (:44-63).
 * d_ohp (n_rows, M) u8 or NULL; d_ohv (n_rows, T) f32 or NULL = 2 * sum_m OHP[i,m] * effects[m,t] (:24). */
int bh_optimal_haploid(bh_ctx *ctx, const uint8_t *d_pop, int64_t n_rows, const int32_t *d_block_start,
                       int64_t n_blocks, uint8_t *d_ohp, float *d_ohv);

/* ---- host-side RNG plumbing (JAX threefry key chains) -------------------------------- */
/* jax.random.key(seed) (vec_env.py:115, Simulator.set_seed). */
void bh_key_from_seed(uint64_t seed, uint32_t key[2]);
/* Sharding windows: so that N shards compute what one device would.  The keys of a call are
 * split for the WHOLE batch and a shard uses its part of them:
 *   env window    this context's envs are envs [env_offset, env_offset + n_envs) of envs_total:
 *                 bh_founders_device and bh_selection_actions use split(env_key, envs_total + 1)[1 + env_offset + e]
 *                 and advance env_key as the whole would; bh_cross* with BH_RNG_INDEPENDENT takes the window in
 *                 crosses (x n_out).  Shared keys need no window: every shard holds the same simulator key.
 *   cross window  one batch of crosses_total crosses cut row-wise (population-sharded): cross b of a call is
 *                 cross cross_offset + b; takes precedence over the env window for bh_cross*.
 * (0, 0) switches a window off (the default). */
int bh_set_env_window(bh_ctx *ctx, int64_t env_offset, int64_t envs_total);
int bh_set_cross_window(bh_ctx *ctx, int64_t cross_offset, int64_t crosses_total);
/* Which threefry layout the device paths of a context follow (bh_cross*, bh_founders_device,
 * bh_selection_actions); default BH_THREEFRY_ORIGINAL.  Masks made ahead are dropped on a change. */
int bh_set_threefry_layout(bh_ctx *ctx, int layout);
/* chromax's mutation probability -- `Simulator(mutation=p)`, which the reference forwards through **kwargs
 * (breedgym/vector/vec_env.py:39,45, breedgym/breedgym.py:74-80).  [EXTERNAL/UNVERIFIED, like the meiosis: chromax
 * is not in the reference tree]  Inside `_meiosis` the gamete key is split into (rec_key, mut_key); every allele of
 * the finished gamete is flipped where uniform(mut_key, (M,)) < p.  0 (the default) is the hot path; with p > 0
 * bh_cross* draw the flips and apply them to every output of the call (bit planes, byte layout), and GEBV / reward
 * are computed from the mutated rows by passes of their own.  Crossover masks and key chains do not change. */
int bh_set_mutation(bh_ctx *ctx, float probability);
/* jax.random.split(key, num) -> out (num, 2); `layout`: bh_threefry_layout. */
int bh_key_split(const uint32_t key[2], int64_t num, int layout, uint32_t *out);
/* jax.random.permutation(key, n) -> out (n) int32. */
int bh_permutation(const uint32_t key[2], int64_t n, int layout, int32_t *out);
/* The founder draw of VecBreedGym.reset (vec_env.py:120-128): keys = split(env_key, E+1);
 * env_key <- keys[0]; env e takes permutation(keys[1+e], N)[:n].  out (E, n) int32 host. */
int bh_founders(uint32_t env_key[2], int64_t n_envs, int64_t n_germplasm, int64_t n, int layout,
                int32_t *out);

/* The same draw made on the device, written to d_out (E, n) int32 in stream order; env_key is advanced on the
 * host.  Germplasm up to 8192 rows: one workgroup per env, Threefry + bitonic sort in LDS; larger (below 2^30
 * rows): only the n wanted positions are followed backwards through JAX's shuffle rounds (two Threefry sweeps per
 * round, nothing sorted), or -- when most of a germplasm of more than 2^18 rows is wanted -- the rounds as stable
 * radix sorts in HBM; envs side by side.  Asynchronous: nothing goes through the host, nothing is waited for.  The
 * scratch of these routes (up to ~1 GB for 2^25 rows; bh_selection_actions shares it) stays with the context until
 * bh_release_scratch. */
int bh_founders_device(bh_ctx *ctx, uint32_t env_key[2], int64_t n_envs, int64_t n_germplasm,
                       int64_t n, int32_t *d_out);

/* bh_founders_device, asked for AHEAD of a bh_cross_packed call: env_key is advanced now, the indices are written
 * by the next two-role launch of the context (its grid grows by n_envs workgroups; for n_germplasm <= 1024) or,
 * failing that, by a launch of their own before anything else of this context runs.  VecBreedGym's horizon step
 * uses it: the reset's founder draw costs no launch of its own between the step and the founder gather. */
int bh_founders_deferred(bh_ctx *ctx, uint32_t env_key[2], int64_t n_envs, int64_t n_germplasm,
                         int64_t n, int32_t *d_out);

#ifdef __cplusplus
}
#endif
#endif

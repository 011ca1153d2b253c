"""``VecBreedGym``: E independent breeding programmes stepped as one fused GPU launch.

Drop-in for ``breedgym.vector.vec_env.VecBreedGym`` (reference ``vec_env.py:30-137``): same
constructor keywords, same ``reset``/``step`` contract, same attributes the wrappers reach
through (``simulator``, ``individual_per_gen``, ``random_key``, ``populations``, ``germplasm``,
``device``, ``step_idx`` ...).  What changes is underneath: ``step`` is one HIP kernel that
gathers the parents through ``actions``, runs meiosis, writes the offspring and accumulates
their GEBV -- ``populations[arange, actions]`` (vec_env.py:89-90) is never materialised and the
GEBV pass (vec_env.py:132-134) never re-reads the offspring.

Differences a caller can see, all opt-in or benign:
  * arrays are ``DeviceArray`` (a torch.Tensor on the GPU that converts to NumPy implicitly,
    like the reference's jax.Array); observations are bool views of the resident rows;
  * ``env_rng="independent"`` gives every env its own crossover randomness; the default
    ``"shared"`` reproduces the reference, whose ``jax.vmap`` over an un-batched simulator key
    (vec_env.py:75-77) hands all envs the same gamete keys;
  * ``sync_rewards=False`` returns the reward vector as a device array and so never blocks;
  * ``storage="bits"`` (the default) keeps the populations between steps as bit-plane rows (1 bit per
    allele, 8x less HBM and 8x less to read per cross).  The observation
    handed out is unchanged -- the same launch also writes the byte-layout offspring -- unless
    ``observations="lazy"``, which returns a ``PackedPopulation`` handle that expands on demand
    (``env.populations``, ``np.asarray(obs)``), for policies that only look at ``infos["GEBV"]``.
    ``storage="bytes"`` keeps the byte layout as the state (the step then reads 4M bytes of parents per
    cross instead of M/2; same results, bit for bit);
  * ``block_pool=16`` (default, with ``storage="bits"``): what ``step`` / ``reset`` hand out -- observation,
    ``infos["GEBV"]``, a device-side reward -- lives in blocks the env keeps and reuses once nothing handed out
    of them is referenced any more (Python reference counts and storage use counts), so a step is one C call with
    nothing to allocate (7 us of host time instead of 19 on the 23 us step of 16 x 256 x 10k) and whatever a caller
    keeps stays valid for as long as it is kept, like the reference's arrays.  At most ``block_pool`` generations
    are kept alive that way; beyond, and with ``block_pool=0``, every step allocates fresh blocks.
"""
from pathlib import Path
from typing import Optional, Tuple, Union

import numpy as np
import torch

from .. import _lib, rng, spaces
from ..engine import DeviceArray, PackedPopulation
from ..simulator import Simulator
from ..utils.paths import DATA_PATH

GENOME_FILE = DATA_PATH.joinpath("small_geno.npy")  # the reference's default (vec_env.py:19)

try:  # a genuine gymnasium VectorEnv when gymnasium is there
    from gymnasium.experimental.vector import VectorEnv as _VectorEnvBase
except ImportError:
    try:
        from gymnasium.vector import VectorEnv as _VectorEnvBase
    except ImportError:
        _VectorEnvBase = object


class VecBreedGym(_VectorEnvBase):
    def __init__(
        self,
        num_envs: int = 1,
        initial_population: Union[str, Path, np.ndarray, torch.Tensor] = GENOME_FILE,
        individual_per_gen: Optional[int] = None,
        num_generations: int = 10,
        autoreset: bool = True,
        reward_shaping: bool = False,
        env_rng: str = "shared",
        sync_rewards: bool = True,
        storage: str = "bits",
        observations: str = "eager",
        block_pool: int = 16,
        **kwargs,
    ):
        if env_rng not in ("shared", "independent"):
            raise ValueError("env_rng must be 'shared' or 'independent'")
        if storage not in ("bytes", "bits"):
            raise ValueError("storage must be 'bytes' or 'bits'")
        if observations not in ("eager", "lazy"):
            raise ValueError("observations must be 'eager' or 'lazy'")
        if observations == "lazy" and storage != "bits":
            raise ValueError("observations='lazy' needs storage='bits'")
        if block_pool == 1 or block_pool < 0:
            raise ValueError("block_pool must be 0 (a fresh block per step) or at least 2")
        self.storage = storage
        self.observations = observations
        self.block_pool = block_pool
        self.num_envs = num_envs
        self.num_generations = num_generations
        self.autoreset = autoreset
        self.reward_shaping = reward_shaping
        self.sync_rewards = sync_rewards
        self._rng_mode = _lib.RNG_SHARED_ENVS if env_rng == "shared" else _lib.RNG_INDEPENDENT
        self.simulator = Simulator(**kwargs)
        self.device = self.simulator.device

        if isinstance(initial_population, (str, Path)):
            if Path(initial_population) == GENOME_FILE and not GENOME_FILE.exists():
                raise FileNotFoundError(
                    f"{GENOME_FILE}: the reference's default population (a 370 x 10k blob its repository strips, "
                    ".MISSING_LARGE_BLOBS) is not shipped here either -- pass initial_population= (a .npy path or an "
                    "(n, M, 2) array; breedgym_amd.synthetic.synthetic_genotypes makes one of the same statistics)"
                )
            germplasm = self.simulator.load_population(initial_population)
        else:
            germplasm = self.simulator.device_put(initial_population)
        self.germplasm = germplasm
        self._germ_rows, _ = self.simulator._rows_of(germplasm)
        if individual_per_gen is None:
            individual_per_gen = len(self.germplasm)
        self.individual_per_gen = individual_per_gen

        self.observation_space = spaces.Box(
            low=0, high=1,
            shape=(self.num_envs, self.individual_per_gen, self.germplasm.shape[1], 2), dtype=np.int8,
        )
        self.action_space = spaces.Box(
            low=0, high=self.individual_per_gen,
            shape=(self.num_envs, self.individual_per_gen, 2), dtype=np.int32,
        )

        self._populations = None
        self._brows = None  # (E * n, pitch) uint8 rows behind `populations`
        self._prows = None  # storage="bits": (E * n, packed_pitch) bit-plane rows, the state between steps
        self._germ_prows = None
        self._n = None
        self._fast_step = hasattr(self.simulator.engine, "cross_step")  # the thin call of the HIP engine
        self._stage, self._stage_turn = {}, 0  # pinned staging blocks for host actions, two per shape, used in turn
        self._ring = None  # StepRing: the output blocks of step(), kept and reused (storage="bits" on the HIP engine)
        self._all_false, self._all_true = np.full(self.num_envs, False), np.full(self.num_envs, True)
        # (indices, key after the draw, bytes of the key drawn from) of the next reset's founders, made by the launch of
        # the step before the horizon (or by the horizon step's, when nothing was made by then)
        self._founders_ahead = None
        self._key_seen = None  # the env key as the last step (or reset) left it, where a draw ahead may follow
        # (slot, what keeps it taken, key after the draw, bytes of the key drawn from): the next reset's founder
        # population, gathered by the horizon step's launch into a slot of the ring
        self._reset_ahead = None
        self._founders_next = None  # the founders of the reset after that one, drawn by the same launch
        self._rew_primed = None  # (E,) f32 the last cross launch left holding -inf: the next step's reward buffer
        self.step_idx = None
        self.reset_infos = {}
        self.random_key = None

    # ---- helpers -----------------------------------------------------------------------------
    @property
    def _rows(self):
        """The byte-layout rows of the current populations (expanded from the bit planes if need be)."""
        if self._brows is None and self._prows is not None:
            self._brows = self.simulator.engine.unpack(self._prows)
        return self._brows

    @property
    def populations(self):
        if self._populations is None and self._rows is not None:
            self._populations = self.simulator.engine.as_genotypes(self._brows, self.num_envs, self._n)
        return self._populations

    @populations.setter
    def populations(self, value):
        # wrappers assign fresh populations (breeding_programs_env.py:56): adopt their rows
        if value is None:
            self._populations, self._brows, self._prows, self._n = None, None, None, None
            return
        rows, lead = self.simulator._rows_of(value)
        if len(lead) != 2 or lead[0] != self.num_envs:
            raise ValueError(f"populations must have shape ({self.num_envs}, n, M, 2)")
        self._adopt(rows, lead[1], self.simulator.engine.pack(rows) if self.storage == "bits" else None)

    def _adopt(self, rows, n, packed=None):
        self._brows, self._prows, self._n = rows, packed, n
        self._populations = None if rows is None else self.simulator.engine.as_genotypes(rows, self.num_envs, n)

    def _observation(self):
        if self.observations == "lazy":
            return PackedPopulation(self.simulator.engine, self._prows, (self.num_envs, self._n))
        return self.populations

    def _device_actions(self, actions) -> torch.Tensor:
        eng = self.simulator.engine
        if isinstance(actions, torch.Tensor):
            a = actions.as_subclass(torch.Tensor)
            if a.device != eng.device or a.dtype != torch.int32 or not a.is_contiguous():
                a = a.to(device=eng.device, dtype=torch.int32).contiguous()
        else:
            a = np.asarray(actions)
            if a.ndim != 3 or a.shape[0] != self.num_envs or a.shape[2] != 2:
                raise ValueError(f"actions must have shape ({self.num_envs}, n, 2), got {a.shape}")
            # host actions (the reference's jax.device_put, vec_env.py:88): converted to int32 straight into a pinned
            # staging block and copied from there without blocking -- a pageable upload costs ~15 us of host time
            stage = self._stage.get(a.shape)
            if stage is None:
                blocks = [torch.empty(a.shape, dtype=torch.int32) for _ in range(2)]
                if eng.device.type == "cuda":
                    blocks = [b.pin_memory() for b in blocks]
                stage = self._stage[a.shape] = [(b, b.numpy(), None) for b in blocks]
            k = self._stage_turn = 1 - self._stage_turn
            pinned, view, ev = stage[k]
            if ev is not None:
                ev.synchronize()  # the copy that last read this block (two steps ago) is long done
            np.copyto(view, a, casting="unsafe")
            a = pinned.to(eng.device, non_blocking=True)
            if eng.device.type == "cuda":
                ev = ev or torch.cuda.Event()
                ev.record()
                stage[k] = (pinned, view, ev)
            return a
        if a.dim() != 3 or a.shape[0] != self.num_envs or a.shape[2] != 2:
            raise ValueError(f"actions must have shape ({self.num_envs}, n, 2), got {tuple(a.shape)}")
        return a

    def cross(self, parents):
        """``vmap(simulator.cross)`` on materialised parents (E, n, 2, M, 2) (vec_env.py:75-77)."""
        E, n = parents.shape[:2]
        M = self.simulator.n_markers
        pool, _ = self.simulator._rows_of(parents.reshape(E * n * 2, M, 2))
        eng = self.simulator.engine
        idx = torch.arange(2 * n, dtype=torch.int32, device=eng.device).reshape(1, n, 2).repeat(E, 1, 1).contiguous()
        out = eng.empty_population(E * n)
        self.simulator.cross_indexed(pool, E, 2 * n, idx, n, out, None, self._rng_mode)
        return eng.as_genotypes(out, E, n)

    def _rewards(self, gebv: torch.Tensor):
        r = self.simulator.engine.reward_max(gebv, self.num_envs)
        if self.sync_rewards:
            return r.cpu().numpy()  # the reference's np.max(...) also syncs here (vec_env.py:97-98)
        return DeviceArray.wrap(r)

    # ---- gymnasium surface -------------------------------------------------------------------
    def step(self, actions) -> Tuple[DeviceArray, np.ndarray, np.ndarray, np.ndarray, dict]:
        if self._n is None:
            raise RuntimeError("call reset() before step()")
        sim = self.simulator
        eng = sim.engine
        if type(actions) is torch.Tensor and actions.dtype == torch.int32 and actions.device == eng.device \
                and actions.dim() == 3 and actions.shape[0] == self.num_envs and actions.shape[2] == 2 \
                and actions.is_contiguous():
            idx = actions  # already what the kernel reads
        else:
            idx = self._device_actions(actions)
        E, n_pool, n_out = self.num_envs, self._n, idx.shape[1]
        done = self.step_idx + 1 == self.num_generations
        want_rew = self.reward_shaping or done
        # this launch also primes the reward buffer of the next step that has a reward
        next_idx = 0 if (done and self.autoreset) else self.step_idx + 1
        want_next = self.reward_shaping or next_idx + 1 == self.num_generations
        ring = self._ring
        if ring is not None and n_out == ring.n_out and self._prows is not None:
            # ---- the usual step: one C call into a block nobody holds any more, nothing allocated (engine.StepRing)
            if self.autoreset and (done or self.step_idx + 2 == self.num_generations):
                # the reset behind the horizon needs its founders: their draw depends on the env key only and rides on
                # a step's launch instead of being a launch of its own (bh_founders_deferred) -- on the launch of the
                # step BEFORE the horizon, so that the horizon step's launch can already gather them: its crosses
                # write no observation, the gathered founders do, and one launch does both instead of two half-empty
                # ones (bh_gather_packed_deferred).  A key somebody advanced in between (SelectionScores does, per
                # step) voids what was made ahead; the draw then rides on the horizon launch and reset() gathers.
                ahead = self._founders_ahead
                if ahead is not None and not (ahead[2] == np.ascontiguousarray(self.random_key, np.uint32).tobytes()
                                              and ahead[0].numel() == E * self.individual_per_gen):
                    ahead = self._founders_ahead = None
                if ahead is None:
                    # a step BEFORE the horizon draws ahead only when the key stood still over the step before it: a
                    # wrapper that advances the env key every step (SelectionScores, the reference's default high-level
                    # wrapper) would void the draw every time -- a draw riding for nothing, a second one on the horizon
                    # launch.  With such a wrapper the draw rides on the horizon launch and reset() gathers.
                    if done or self._key_seen == np.ascontiguousarray(self.random_key, np.uint32).tobytes():
                        self._founders_ahead = self._draw_founders(deferred=True)
                elif done and self._ring is not None and self._germ_prows is not None:
                    # ... and the same launch draws the founders of the reset AFTER that one (from the key that draw
                    # left): the next horizon finds them made, no step in between carries a draw
                    nxt = self._draw_founders(deferred=True, key=ahead[1])
                    self._reset_ahead = self._gather_ahead(ahead)
                    if self._reset_ahead is not None:
                        self._founders_next = nxt
            # at the horizon the observation handed out is the next episode's first one (vec_env.py:102-107):
            # this generation's byte layout would be written for nobody
            want_obs = self.observations == "eager" and not (done and self.autoreset)
            key = sim.random_key
            if key.dtype != np.uint32 or not key.flags.c_contiguous:
                key = sim.random_key = np.ascontiguousarray(key, np.uint32)
            got = ring.step(self._prows.data_ptr(), n_pool, idx, key, self._rng_mode | (ring.OBS if want_obs else 0),
                            want_rew, want_next)
            if got is not None:
                slot, rslot = got
                self._prows, self._n = ring.state[slot], n_out
                self._brows = ring.obs_rows[slot] if want_obs else None
                self._populations = ring.populations[slot] if want_obs else None
                self.step_idx += 1
                if self.autoreset and self.step_idx + 2 >= self.num_generations:
                    self._key_seen = np.ascontiguousarray(self.random_key, np.uint32).tobytes()  # (see the draw ahead, above)
                infos = {"GEBV": ring.gebv[slot]}
                if rslot is None:
                    rews = np.zeros(E)
                elif self.sync_rewards:
                    rews = ring.reward[rslot].cpu().numpy()  # the reference's np.max(...) also syncs here (vec_env.py:97-98)
                else:
                    rews = ring.reward[rslot]
                if done and self.autoreset:
                    obs, _ = self.reset()  # like the reference: obs is post-reset, infos pre-reset (vec_env.py:102-107)
                else:
                    obs = self._populations if self.observations == "eager" else ring.lazy[slot]
                return obs, rews, self._all_false.copy(), (self._all_true if done else self._all_false).copy(), infos
            # every block is held by somebody and the pool may not grow: fresh blocks for this step
            return self._step_allocating(idx, n_pool, n_out, done, want_rew, want_next, founders_drawn=True)
        return self._step_allocating(idx, n_pool, n_out, done, want_rew, want_next)

    def _step_allocating(self, idx, n_pool, n_out, done, want_rew, want_next, founders_drawn=False):
        """``step`` with a fresh block per output (``ring_depth=0``, ``storage="bytes"``, a generation whose size
        differs from the ring's, an engine without the ring)."""
        sim, E = self.simulator, self.num_envs
        eng = sim.engine
        # the reward (max GEBV per env, vec_env.py:96-98) comes out of the same launch when it is due: the row
        # waves note their GEBV with atomic maxima into a buffer the launch before left holding -inf
        rew_dev = None
        if want_rew:
            rew_dev, self._rew_primed = self._rew_primed, None
            if rew_dev is None or rew_dev.shape[0] != E:
                rew_dev = eng.empty_reward(E)  # not primed: the library fills it first
        if self.storage == "bits":
            if done and self.autoreset and not founders_drawn:
                self._founders_ahead = self._draw_founders(deferred=True)
            want_obs = self.observations == "eager" and not (done and self.autoreset)
            if self._fast_step:
                key = sim.random_key
                if key.dtype != np.uint32 or not key.flags.c_contiguous:
                    key = sim.random_key = np.ascontiguousarray(key, np.uint32)
                outp, out, gebv, rew_next = eng.cross_step(self._prows, E, n_pool, idx, n_out, key, self._rng_mode,
                                                           want_obs, rew_dev, want_next)
            else:
                gebv = eng.empty((E, n_out, eng.n_traits), torch.float32)
                rew_next = eng.empty_reward(E) if want_next else None
                outp = eng.empty_packed(E * n_out)
                out = eng.empty_population(E * n_out) if want_obs else None
                sim.cross_indexed_packed(self._prows, E, n_pool, idx, n_out, outp, out, gebv, self._rng_mode, rew_dev,
                                         rew_next)
            self._adopt(out, n_out, outp)
        else:
            gebv = eng.empty((E, n_out, eng.n_traits), torch.float32)
            rew_next = eng.empty_reward(E) if want_next else None
            out = eng.empty_population(E * n_out)
            sim.cross_indexed(self._brows, E, n_pool, idx, n_out, out, gebv, self._rng_mode, rew_dev, rew_next)
            self._adopt(out, n_out)
        self._rew_primed = rew_next
        return self._close_step({"GEBV": gebv.as_subclass(DeviceArray)}, rew_dev)

    def _draw_founders(self, deferred=False, key=None):
        """The founder draw of a reset from the env key as it stands (or ``key``) -> (indices, key after, bytes of the
        key drawn from)."""
        key = np.ascontiguousarray(self.random_key if key is None else key, np.uint32)
        idx, new_key = self.simulator.engine.founders(key, self.num_envs, len(self.germplasm), self.individual_per_gen,
                                                      deferred=deferred)
        return idx, new_key, key.tobytes()

    def _gather_ahead(self, ahead):
        """The founders of the next reset (drawn already: ``ahead``) gathered into a slot of the ring by the launch of
        the cross call that follows -> what ``reset`` needs to adopt them, or None (no slot to be had)."""
        ring, eng = self._ring, self.simulator.engine
        E, n = self.num_envs, self.individual_per_gen
        if ring is None or (ring.n_envs, ring.n_out) != (E, n) or self._germ_prows is None:
            return None
        slot = ring.take()
        if slot is None:
            return None
        # (the founders' GEBV goes to a block of its own, not the slot's: reset_infos keeps it until the next reset, and
        # holding the slot's view that long would keep the slot out of use)
        gebv = eng.empty((E, n, eng.n_traits), torch.float32).as_subclass(DeviceArray)
        hold = (ring.state[slot], gebv, ahead[0])  # a slot is taken for as long as something of it is referenced
        eng.gather_packed(self._germ_prows, ahead[0], E * n, ring.state[slot], ring.obs_rows[slot], gebv, deferred=True)
        return slot, hold, ahead[0], ahead[1], ahead[2]

    def _close_step(self, infos, rew_dev=None):
        """What follows the new generation in any kind of step (vec_env.py:92-107): the generation is counted, the
        reward -- max GEBV per env -- is handed out when it is due (from the launch that noted it, ``rew_dev``, or from a
        reduction over ``infos["GEBV"]``), the horizon resets the envs, and the step's five values go back: the
        observation is the post-reset one at the horizon, the infos are the pre-reset ones."""
        self.step_idx += 1
        done = self.step_idx == self.num_generations
        if not (self.reward_shaping or done):
            rews = np.zeros(self.num_envs)
        elif rew_dev is None:
            rews = self._rewards(infos["GEBV"].as_subclass(torch.Tensor))
        elif self.sync_rewards:
            rews = rew_dev.cpu().numpy()  # the reference's np.max(...) also syncs here (vec_env.py:97-98)
        else:
            rews = rew_dev.as_subclass(DeviceArray)
        obs = self.reset()[0] if done and self.autoreset else self._observation()
        return obs, rews, self._all_false.copy(), (self._all_true if done else self._all_false).copy(), infos

    def reset(self, seed: Optional[int] = None, options: Optional[dict] = None) -> Tuple[DeviceArray, dict]:
        self.step_idx = 0
        if seed is not None:
            self.simulator.set_seed(seed)
            self.random_key = rng.key(seed)
        elif self.random_key is None:
            seed = int(np.random.randint(2**32, dtype=np.uint64))
            self.random_key = rng.key(seed)

        if options is not None and "individual_per_gen" in options.keys():
            self.individual_per_gen = options["individual_per_gen"]

        eng = self.simulator.engine
        E, n = self.num_envs, self.individual_per_gen
        ahead, self._founders_ahead = self._founders_ahead, None
        made, self._reset_ahead = self._reset_ahead, None
        key_now = np.ascontiguousarray(self.random_key, np.uint32).tobytes()
        if (made is not None and seed is None and self._ring is not None and made[4] == key_now
                and (self._ring.n_envs, self._ring.n_out) == (E, n) and self.storage == "bits"):
            # the horizon step's launch gathered this very population already (_gather_ahead)
            slot, hold, _, self.random_key, _ = made
            self._key_seen = np.ascontiguousarray(self.random_key, np.uint32).tobytes()
            self._founders_ahead, self._founders_next = self._founders_next, None
            ring = self._ring
            self._prows, self._brows, self._populations, self._n = (ring.state[slot], ring.obs_rows[slot],
                                                                    ring.populations[slot], n)
            self._rew_primed = None
            self.reset_infos = {"GEBV": hold[1]}
            return (self._populations if self.observations == "eager" else ring.lazy[slot]), self.reset_infos
        del made
        self._founders_next = None
        if ahead is not None and seed is None and ahead[0].numel() == E * n and ahead[2] == key_now:
            idx, self.random_key = ahead[0], ahead[1]  # drawn by a step's launch from this same key
        else:
            idx, self.random_key = eng.founders(self.random_key, E, len(self.germplasm), n)
        self._key_seen = np.ascontiguousarray(self.random_key, np.uint32).tobytes()
        if self.storage == "bits":
            if self._germ_prows is None:
                self._germ_prows = eng.pack(self._germ_rows)
            if self.block_pool and hasattr(eng, "step_ring") and (
                    self._ring is None or (self._ring.n_envs, self._ring.n_out) != (E, n)):
                if self._ring is not None:
                    self._ring.close()
                self._ring = eng.step_ring(E, n, self.observations == "eager", self.block_pool)
                if self._ring is None:  # this torch build cannot tell who holds a block (engine.block_pool_supported)
                    self.block_pool = 0
            slot = self._ring.take() if self._ring is not None else None
            if slot is not None:
                ring = self._ring
                # (a block of its own, not the slot's: holding the slot's GEBV view until the next reset would keep the
                # slot out of use)
                gebv = eng.empty((E, n, eng.n_traits), torch.float32).as_subclass(DeviceArray)
                eng.gather_packed(self._germ_prows, idx, E * n, ring.state[slot], ring.obs_rows[slot], gebv)
                self._prows, self._brows, self._populations, self._n = (ring.state[slot], ring.obs_rows[slot],
                                                                        ring.populations[slot], n)
                self._rew_primed = None
                self.reset_infos = {"GEBV": gebv}
                return (self._populations if self.observations == "eager" else ring.lazy[slot]), self.reset_infos
            gebv = eng.empty((E, n, eng.n_traits), torch.float32)
            prows = eng.empty_packed(E * n)
            rows = eng.empty_population(E * n) if self.observations == "eager" else None
            eng.gather_packed(self._germ_prows, idx, E * n, prows, rows, gebv)
            self._adopt(rows, n, prows)
        else:
            gebv = eng.empty((E, n, eng.n_traits), torch.float32)
            rows = eng.empty_population(E * n)
            eng.gather(self._germ_rows, idx, rows, gebv)
            self._adopt(rows, n)
        self.reset_infos = {"GEBV": DeviceArray.wrap(gebv)}
        return self._observation(), self.reset_infos

    def get_info(self) -> dict:
        if self._brows is None and self._prows is not None:  # GEBV straight from the bit planes
            eng = self.simulator.engine
            gebv = eng.empty((self.num_envs, self._n, eng.n_traits), torch.float32)
            eng.gather_packed(self._prows, None, self.num_envs * self._n, None, None, gebv)
            return {"GEBV": DeviceArray.wrap(gebv)}
        return {"GEBV": self.simulator.GEBV_model(self.populations)}

    def set_attr(self, name, values):
        return setattr(self, name, values)

    # ---- checkpoint / resume ---------------------------------------------------------------------
    # The reference has none; its env state is `populations`, `step_idx`, `random_key` and the simulator's key
    # (vec_env.py:70-73).  Restoring those four continues an episode bit for bit.
    def get_state(self) -> dict:
        """Everything a later ``set_state`` needs to continue exactly here, as host arrays (populations as bit planes:
        1 bit per allele)."""
        if self._n is None:
            raise RuntimeError("call reset() before get_state()")
        eng = self.simulator.engine
        packed = self._prows if self._prows is not None else eng.pack(self._brows)
        return {
            "populations_packed": packed.cpu().numpy().copy(), "individual_per_gen": int(self._n),
            "num_envs": self.num_envs, "n_markers": int(self.simulator.n_markers), "step_idx": int(self.step_idx),
            "random_key": np.array(self.random_key, np.uint32).copy(),
            "simulator_key": np.array(self.simulator.random_key, np.uint32).copy(),
        }

    def set_state(self, state: dict) -> None:
        eng = self.simulator.engine
        if state["num_envs"] != self.num_envs or state["n_markers"] != self.simulator.n_markers:
            raise ValueError("the state belongs to an env of another shape")
        n = int(state["individual_per_gen"])
        packed = torch.from_numpy(np.ascontiguousarray(state["populations_packed"])).to(eng.device)
        if packed.shape != (self.num_envs * n, eng.packed_pitch):
            raise ValueError("populations_packed has the wrong shape")
        rows = eng.unpack(packed) if (self.storage == "bytes" or self.observations == "eager") else None
        self._adopt(rows, n, packed if self.storage == "bits" else None)
        self.step_idx = int(state["step_idx"])
        self.random_key = np.array(state["random_key"], np.uint32).copy()
        self.simulator.random_key = np.array(state["simulator_key"], np.uint32).copy()
        self._founders_ahead, self._founders_next, self._reset_ahead, self._rew_primed = None, None, None, None

    def close(self, **kwargs):
        if self._ring is not None:
            self._ring.close()
            self._ring = None

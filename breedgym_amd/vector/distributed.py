"""Multi-GPU: one process per GPU, ``torch.distributed`` (RCCL over xGMI on MI355X, gloo on CPU).

Two ways the hot path shards:

* **env-sharded** -- ``DistributedBreedGym``.  The role of the reference's
  ``DistributedBreedGym`` (``vec_env.py:150-236``: one worker process per device, observations
  piped back to a parent through Gymnasium's ``AsyncVectorEnv``).  Here every rank simply owns
  ``envs_per_device`` envs of a ``VecBreedGym`` on its own GPU; ``step`` has no collective at
  all, observations never leave the device, and only what a learner on another rank needs
  (rewards, GEBV) is all-gathered on request -- 4 B per env, 4 B x n x T per env.
* **population-sharded** -- ``ShardedPopulation``.  One large population (BASELINE config 4:
  65,536 individuals x 100k markers) split row-wise over the ranks.  Per generation:
  local GEBV -> local top-k -> all-gather of (score, global index) candidates (8 B x k per rank)
  -> every rank computes the same global top-k -> the winners' genotype rows are replicated with ONE
  all-gather: every rank sends the winners it owns, padded to the largest share (the global index list is
  identical everywhere, so every rank derives all counts and offsets from it -- nothing but the rows is
  exchanged; (N-1)/N of the rows per rank on the wire, half of what an all-reduce of a zero-padded buffer
  moves) -> each rank crosses its own shard of the next generation straight from the gathered buffer
  (parent indices are remapped, rows are never permuted).  The scalar best-GEBV reward is a MAX all-reduce.

Both are **world-size invariant**: N ranks compute, shard by shard, exactly what one device would
compute for the whole -- the engine's sharding windows (include/breedhip.h) make every rank draw the
keys of the whole batch and use its own part of them, and all ranks start from the same seed.

Nothing here launches kernels of its own: compute goes through the engine of the wrapped env.
"""
from typing import Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist

from ..engine import DeviceArray
from .vec_env import VecBreedGym


def _world(group=None) -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


class DistributedBreedGym:
    """``num_envs = envs_per_device * world_size`` breeding programmes, ``envs_per_device`` per rank."""

    def __init__(self, envs_per_device: int, initial_population, group=None, **kwargs):
        self.group = group
        self.rank, self.world_size = _world(group)
        self.envs_per_device = envs_per_device
        self.num_envs = envs_per_device * self.world_size
        self.env = VecBreedGym(num_envs=envs_per_device, initial_population=initial_population, **kwargs)
        self.device = self.env.device
        # env e of this rank is env rank * envs_per_device + e of num_envs: founder draws, selection shuffles and
        # independent crossover keys are those a single device would use for it
        self.env.simulator.engine.set_env_window(self.rank * envs_per_device, self.num_envs)

    # the slice of a global (num_envs, ...) array this rank owns
    def local_slice(self) -> slice:
        return slice(self.rank * self.envs_per_device, (self.rank + 1) * self.envs_per_device)

    def reset(self, seed: Optional[int] = None, options: Optional[dict] = None):
        """The same seed on every rank: the key chains are those of one ``VecBreedGym`` of ``num_envs`` envs, each
        rank draws from them through its env window.  (seed=None on first use: rank 0's choice is broadcast.)"""
        if seed is None and self.env.random_key is None:
            t = torch.randint(0, 2**31 - 1, (1,), dtype=torch.int64)
            if self.world_size > 1:
                t = t.to(self.device) if self.device.type == "cuda" else t
                dist.broadcast(t, src=0, group=self.group)
            seed = int(t.item())
        return self.env.reset(seed=seed, options=options)

    def step(self, actions):
        """`actions`: this rank's (envs_per_device, n, 2) slice, or the global (num_envs, n, 2) array."""
        if actions.shape[0] == self.num_envs and self.world_size > 1:
            actions = actions[self.local_slice()]
        return self.env.step(actions)

    def all_gather(self, local: torch.Tensor) -> torch.Tensor:
        """(envs_per_device, ...) on every rank -> (num_envs, ...) on every rank."""
        t = local.as_subclass(torch.Tensor).contiguous()
        if self.world_size == 1:
            return t
        out = torch.empty((self.world_size, *t.shape), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t, group=self.group) if t.is_cuda else dist.all_gather(
            list(out.unbind(0)), t, group=self.group
        )
        return out.reshape(self.num_envs, *t.shape[1:])

    def gather_rewards(self, local_rewards) -> np.ndarray:
        t = local_rewards if isinstance(local_rewards, torch.Tensor) else torch.as_tensor(np.asarray(local_rewards))
        t = t.to(device=self.device, dtype=torch.float32)
        return self.all_gather(t).cpu().numpy()

    def best_reward(self, local_rewards) -> float:
        """max over all envs of all ranks: a MAX all-reduce of one float."""
        t = local_rewards if isinstance(local_rewards, torch.Tensor) else torch.as_tensor(np.asarray(local_rewards))
        m = t.to(device=self.device, dtype=torch.float32).as_subclass(torch.Tensor).max().reshape(1)
        if self.world_size > 1:
            dist.all_reduce(m, op=dist.ReduceOp.MAX, group=self.group)
        return float(m.item())

    def close(self):
        self.env.close()


class ShardedPopulation:
    """One population of ``n_total`` individuals, rows ``[rank * n_local, (rank + 1) * n_local)`` here.

    ``storage="bits"`` keeps the shard -- and, more to the point, the winners' rows every generation moves
    between the ranks -- as bit-plane rows: 8x less HBM and 8x less on the xGMI links (at BASELINE config 4:
    164 MB per generation instead of 1.31 GB).

    World-size invariant: with the same ``key`` on every rank and ``pairs`` = the rank's rows of one global
    (n_total, 2) pair table, the N-rank generation is the 1-rank generation of the whole population, row for
    row (the engine's cross window makes rank r use keys ``1 + 2 (r n_local + b) + p`` of one
    ``split(key, 2 n_total + 1)``)."""

    def __init__(self, simulator, local_rows: torch.Tensor, group=None, storage: str = "bytes"):
        if storage not in ("bytes", "bits"):
            raise ValueError("storage must be 'bytes' or 'bits'")
        self.sim = simulator
        self.eng = simulator.engine
        self.group = group
        self.storage = storage
        self.rank, self.world_size = _world(group)
        # (n_local, pitch) byte-layout rows, or (n_local, packed_pitch) bit-plane rows, on this rank's device
        self._store = self.eng.pack(local_rows) if storage == "bits" else local_rows
        self.n_local = local_rows.shape[0]
        self.n_total = self.n_local * self.world_size

    @property
    def rows(self) -> torch.Tensor:
        """This rank's shard in the byte layout (expanded from the bit planes if need be)."""
        return self.eng.unpack(self._store) if self.storage == "bits" else self._store

    def _all_gather(self, t: torch.Tensor) -> torch.Tensor:
        if self.world_size == 1:
            return t.unsqueeze(0)
        out = torch.empty((self.world_size, *t.shape), dtype=t.dtype, device=t.device)
        if t.is_cuda:
            dist.all_gather_into_tensor(out, t.contiguous(), group=self.group)
        else:
            dist.all_gather(list(out.unbind(0)), t.contiguous(), group=self.group)
        return out

    def _gather_local(self, src, idx, out) -> None:
        if self.storage == "bits":
            self.eng.gather_packed(src, idx, idx.numel(), out)
        else:
            self.eng.gather(src, idx, out)

    def gebv(self) -> torch.Tensor:
        if self.storage == "bits":
            g = self.eng.empty((self.n_local, self.eng.n_traits), torch.float32)
            self.eng.gather_packed(self._store, None, self.n_local, None, None, g)
            return g
        return self.eng.gebv(self._store)

    def global_topk(self, local_scores: torch.Tensor, k: int) -> torch.Tensor:
        """Global indices (descending score, ties to the lower global index) of the k best of the
        whole population, identical on every rank.  Candidates: each rank's local top-min(k, n_local)."""
        s = local_scores.as_subclass(torch.Tensor).reshape(-1).to(torch.float32)
        kk = min(k, self.n_local)
        loc = self.eng.topk(s, kk).to(torch.int64)
        cand_s = self._all_gather(s[loc])  # (W, kk)
        cand_i = self._all_gather(loc + self.rank * self.n_local)  # (W, kk) global indices
        # jax.lax.top_k on the concatenated scores resolves ties to the lower index.  The candidate list is
        # rank-major and, inside a rank, in the local top-k's order (ties already to the lower index), ranks own
        # ascending index ranges: the earlier position IS the lower global index, so a top-k with ties to the
        # earlier position needs no re-ordering of the candidates
        best = self.eng.topk(cand_s.reshape(-1), k).to(torch.int64)
        return cand_i.reshape(-1)[best]

    def replicate_winners(self, global_idx: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """The rows of ``global_idx`` (k,) on every rank: ``(pool, pos)`` with winner j in row ``pos[j]`` of
        ``pool`` (rows in the storage layout).

        Winners are owned by contiguous index ranges, and ``global_idx`` is identical on all ranks: every rank
        counts, from that list alone, how many winners each rank owns.  Each rank gathers ITS winners (in list
        order) into a send block padded to the largest share and ONE all-gather of those blocks replicates
        them: (N-1)/N of the rows per rank on the wire.  The blocks are not re-ordered afterwards -- ``pos``
        says where a winner landed -- so the only extra traffic is the padding (top-k winners spread evenly:
        a few per cent).  One small device-to-host copy (the N counts) sizes the blocks."""
        k = global_idx.numel()
        if self.world_size == 1:
            buf = torch.empty((k, self._store.shape[1]), dtype=torch.uint8, device=self._store.device)
            self._gather_local(self._store, global_idx.to(torch.int32).contiguous(), buf)
            return buf, torch.arange(k, device=buf.device)
        owner = torch.div(global_idx, self.n_local, rounding_mode="floor")
        hot = torch.nn.functional.one_hot(owner, self.world_size)  # (k, W)
        within = (hot.cumsum(0) - hot).gather(1, owner.unsqueeze(1)).squeeze(1)  # earlier winners of the same owner
        counts = hot.sum(0).cpu()
        cmax = int(counts.max())
        mine = owner == self.rank
        send_idx = torch.zeros((cmax,), dtype=torch.int32, device=self._store.device)  # padding: row 0 again
        send_idx[: int(counts[self.rank])] = (global_idx[mine] - self.rank * self.n_local).to(torch.int32)
        block = torch.empty((cmax, self._store.shape[1]), dtype=torch.uint8, device=self._store.device)
        self._gather_local(self._store, send_idx, block)
        pool = self._all_gather(block).reshape(self.world_size * cmax, -1)
        return pool, owner * cmax + within

    def gather_rows(self, global_idx: torch.Tensor) -> torch.Tensor:
        """The genotype rows of `global_idx` (k,), in that order, replicated on every rank (byte layout)."""
        pool, pos = self.replicate_winners(global_idx)
        rows = pool.index_select(0, pos)
        return self.eng.unpack(rows) if self.storage == "bits" else rows

    def select_and_cross(self, k: int, pairs_local: torch.Tensor, key: np.ndarray) -> DeviceArray:
        """One generation: global top-k by GEBV (first trait), replicate the winners, cross this rank's
        shard of the next generation.  ``pairs_local`` (n_local, 2) int32 indexes the k winners (best first);
        ``key`` is the simulator key, THE SAME on every rank, advanced in place as one device would."""
        scores = self.gebv()[:, 0]
        best = self.global_topk(scores, k)
        pool, pos = self.replicate_winners(best)
        gebv = self.eng.empty((self.n_local, self.eng.n_traits), torch.float32)
        # winner j sits in row pos[j] of the gathered buffer: point the pairs there instead of moving rows
        idx = pos[pairs_local.to(device=pool.device, dtype=torch.int64)].to(torch.int32).reshape(1, -1, 2).contiguous()
        # this rank's crosses are crosses [rank * n_local, ...) of ONE batch of n_total: the window is set for
        # this call only (the simulator's engine is shared with whoever else crosses on it)
        self.eng.set_cross_window(self.rank * self.n_local, self.n_total)
        try:
            if self.storage == "bits":
                out = self.eng.empty_packed(self.n_local)
                self.eng.cross_packed(pool, 1, pool.shape[0], idx, self.n_local, key, 1, out, None, gebv)
            else:
                out = self.eng.empty_population(self.n_local)
                self.eng.cross(pool, 1, pool.shape[0], idx, self.n_local, key, 1, out, gebv)
        finally:
            self.eng.set_cross_window(0, 0)
        self._store = out
        return DeviceArray.wrap(gebv)

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
import contextlib
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


def _device_collectives(group=None) -> bool:
    """Whether the process group moves device tensors itself (RCCL).  Under gloo -- the CPU tests, and the rehearsal
    of several ranks on ONE GPU, where RCCL refuses two ranks per device -- device tensors are staged through the
    host: slow, correct, and never the measured configuration."""
    return dist.get_backend(group) == "nccl"


def all_gather_stacked(t: torch.Tensor, world_size: int, group=None) -> torch.Tensor:
    """(...) on every rank -> (world_size, ...) on every rank, on t's device."""
    t = t.as_subclass(torch.Tensor).contiguous()
    if world_size == 1:
        return t.unsqueeze(0)
    if t.is_cuda and _device_collectives(group):
        out = torch.empty((world_size, *t.shape), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t, group=group)
        return out
    host = t.cpu()
    parts = [torch.empty_like(host) for _ in range(world_size)]
    dist.all_gather(parts, host, group=group)
    return torch.stack(parts).to(t.device)


def _reduce_small(t: torch.Tensor, op, group=None) -> torch.Tensor:
    """all-reduce of a few numbers, in place on t's device (staged through the host under gloo)."""
    if t.is_cuda and not _device_collectives(group):
        h = t.cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op, group=group)
    return t


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
            if self.world_size > 1:  # a host tensor under gloo, a device tensor under RCCL
                t = t.to(self.device) if self.device.type == "cuda" and _device_collectives(self.group) else t
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
        return all_gather_stacked(t, self.world_size, self.group).reshape(self.num_envs, *t.shape[1:])

    def gather_rewards(self, local_rewards) -> np.ndarray:
        t = local_rewards if isinstance(local_rewards, torch.Tensor) else torch.as_tensor(np.asarray(local_rewards))
        t = t.to(device=self.device, dtype=torch.float32)
        return self.all_gather(t).cpu().numpy()

    def best_reward(self, local_rewards) -> float:
        """max over all envs of all ranks: a MAX all-reduce of one float."""
        t = local_rewards if isinstance(local_rewards, torch.Tensor) else torch.as_tensor(np.asarray(local_rewards))
        m = t.to(device=self.device, dtype=torch.float32).as_subclass(torch.Tensor).max().reshape(1)
        if self.world_size > 1:
            _reduce_small(m, dist.ReduceOp.MAX, self.group)
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
    ``split(key, 2 n_total + 1)``).

    No host round trip on the data path of a generation: where winners live and land is worked out on the device
    (``bh_winner_layout``), and the one number the host needs -- how many rows the largest share has, which sizes the
    all-gather -- is *speculated* from the generation before (``cap_slack`` above the last largest share), read back
    from pinned memory only after the generation's cross has been queued, and the generation is redone with the
    exact size in the (rare, tested) case the guess was too small.  The first generation sizes its blocks exactly."""

    def __init__(self, simulator, local_rows: torch.Tensor, group=None, storage: str = "bytes", cap_slack: float = 0.125):
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
        self.cap_slack = float(cap_slack)
        self._cap_hint = None  # rows per block of the winners' all-gather, guessed for the next generation
        self._cap_k = None     # ... and the k it was guessed for
        self.redone = 0        # generations whose guess was too small
        self._gebv_of_store = None  # (n_local, T) the GEBV of `_store` where the cross that wrote it took it
        cuda = self._store.is_cuda
        self._counts_host = torch.empty((self.world_size + 1,), dtype=torch.int32, pin_memory=cuda)
        self._counts_ready = torch.cuda.Event() if cuda else None

    @property
    def rows(self) -> torch.Tensor:
        """This rank's shard in the byte layout (expanded from the bit planes if need be)."""
        return self.eng.unpack(self._store) if self.storage == "bits" else self._store

    def _all_gather(self, t: torch.Tensor) -> torch.Tensor:
        return all_gather_stacked(t, self.world_size, self.group)

    def _gather_local(self, src, idx, out) -> None:
        if self.storage == "bits":
            self.eng.gather_packed(src, idx, idx.numel(), out)
        else:
            self.eng.gather(src, idx, out)

    def gebv(self) -> torch.Tensor:
        # the GEBV of the shard as the cross that made it reported it (select_and_cross): the same rows, so the pass over
        # them is not made a second time (63 us of a 1.78 ms generation at 8192 x 100k)
        if self._gebv_of_store is not None:
            return self._gebv_of_store
        if self.storage == "bits":
            g = self.eng.empty((self.n_local, self.eng.n_traits), torch.float32)
            self.eng.gather_packed(self._store, None, self.n_local, None, None, g)
            return g
        return self.eng.gebv(self._store)

    def global_topk(self, local_scores: torch.Tensor, k: int) -> torch.Tensor:
        """Global indices (k,) int32 (descending score, ties to the lower global index) of the k best of the
        whole population, identical on every rank.  Candidates: each rank's local top-min(k, n_local), exchanged
        with ONE all-gather of (score bits, global index) pairs."""
        s = local_scores.as_subclass(torch.Tensor).reshape(-1).to(device=self._store.device, dtype=torch.float32)
        kk = min(k, self.n_local)
        loc = self.eng.topk(s, kk)
        cand = torch.stack((s[loc.to(torch.int64)].view(torch.int32), loc + self.rank * self.n_local), dim=1)  # (kk, 2)
        cand = self._all_gather(cand).reshape(-1, 2)  # (W kk, 2), rank-major
        # jax.lax.top_k on the concatenated scores resolves ties to the lower index.  The candidate list is
        # rank-major and, inside a rank, in the local top-k's order (ties already to the lower index), ranks own
        # ascending index ranges: the earlier position IS the lower global index, so a top-k with ties to the
        # earlier position needs no re-ordering of the candidates
        best = self.eng.topk(cand[:, 0].contiguous().view(torch.float32), k).to(torch.int64)
        return cand[:, 1][best].contiguous()

    # ---- the winners' rows on every rank --------------------------------------------------------------------------
    def _counts_to_host(self, counts: torch.Tensor) -> None:
        if self._counts_ready is not None:
            # the copy and its event go where the layout kernel went: the ENGINE's stream (bh_set_stream), which need
            # not be torch's current one -- on another stream the host could read the counts before they are written
            # and accept blocks that are too small
            es = getattr(self.eng, "_stream", None)
            with torch.cuda.stream(es) if es is not None else contextlib.nullcontext():
                self._counts_host.copy_(counts, non_blocking=True)
                self._counts_ready.record()
        else:
            self._counts_host.copy_(counts)

    def _largest_share(self) -> int:
        """The largest number of winners one rank owns, as the last layout call found it (waits for that call only)."""
        if self._counts_ready is not None:
            self._counts_ready.synchronize()
        return int(self._counts_host[self.world_size])

    def _cap_bound(self, k: int) -> int:
        return max(1, min(k, self.n_local))

    def _exact_cap(self, winners: torch.Tensor) -> int:
        """The exact block size, at the price of one wait for the device (first generation, and after a bad guess)."""
        if self.world_size == 1:  # one rank owns every winner: nothing to count, nothing to wait for
            return self._cap_bound(winners.numel())
        _, _, counts, _ = self.eng.winner_layout(winners, self.n_local, self.world_size, self.rank, 1)
        self._counts_to_host(counts)
        return max(1, self._largest_share())

    def _replicate(self, winners: torch.Tensor, cap: int, pairs: Optional[torch.Tensor] = None):
        """Queue: layout -> this rank's block of `cap` rows -> one all-gather.  -> (pool (W cap, row bytes), pos (k,),
        pairs remapped to pool rows or None).  Valid iff ``_largest_share() <= cap`` afterwards."""
        send_idx, pos, counts, idx = self.eng.winner_layout(winners, self.n_local, self.world_size, self.rank, cap, pairs)
        self._counts_to_host(counts)
        block = torch.empty((cap, self._store.shape[1]), dtype=torch.uint8, device=self._store.device)
        self._gather_local(self._store, send_idx, block)
        pool = self._all_gather(block).reshape(self.world_size * cap, -1)
        return pool, pos, idx

    def _next_cap(self, k: int, largest: int) -> int:
        return min(self._cap_bound(k), largest + int(largest * self.cap_slack) + 16)

    def replicate_winners(self, global_idx: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """The rows of ``global_idx`` (k,) on every rank: ``(pool, pos)`` with winner j in row ``pos[j]`` of
        ``pool`` (rows in the storage layout).

        Winners are owned by contiguous index ranges, and ``global_idx`` is identical on all ranks: every rank
        derives, from that list alone, how many winners each rank owns and where each lands.  Each rank gathers ITS
        winners (in list order) into a block padded to the largest share and ONE all-gather of those blocks replicates
        them: (N-1)/N of the rows per rank on the wire.  The blocks are not re-ordered afterwards -- ``pos`` says
        where a winner landed.  (This stand-alone form sizes its blocks exactly and so waits for the device once;
        ``select_and_cross`` speculates instead.)"""
        w = global_idx.as_subclass(torch.Tensor).to(torch.int32).contiguous()
        cap = self._exact_cap(w)
        pool, pos, _ = self._replicate(w, cap)
        return pool, pos.to(torch.int64)

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
        pairs = pairs_local.as_subclass(torch.Tensor).to(device=self._store.device, dtype=torch.int32).reshape(1, -1, 2).contiguous()
        if self._cap_k != k:  # a guess made for another k says nothing about this one
            self._cap_hint, self._cap_k = None, k
        cap = self._cap_hint if self._cap_hint is not None else self._exact_cap(best)
        key_before = key.copy()
        while True:
            # winner j sits in row pos[j] of the gathered buffer: the pairs are pointed there, rows are never moved
            pool, _, idx = self._replicate(best, cap, pairs)
            gebv = self.eng.empty((self.n_local, self.eng.n_traits), torch.float32)
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
            # only now does the host look at the count the layout found: the device has the whole generation queued
            largest = self._largest_share()
            if largest <= cap:
                break
            self.redone += 1  # the guess was too small: some winners had no row.  Same key, exact size, again
            key[:] = key_before
            cap = largest
        self._cap_hint = self._next_cap(k, largest)
        self._store = out
        self._gebv_of_store = gebv
        return DeviceArray.wrap(gebv)

"""Test double: an engine whose operators are the CPU oracle.

TEST INFRASTRUCTURE ONLY.  It lets the CPU test-suite drive the *host logic* of the product
(``Simulator``, ``VecBreedGym``, ``BreedGym``, the sharding driver) without a GPU, through the
``engine=`` seam of ``Simulator``.  Nothing in ``breedgym_amd`` imports this.
"""
import numpy as np
import torch

from breedgym_amd.engine import EngineBase
from oracle import c_oracle as co


def packed_pitch_of(n_markers: int) -> int:
    """Bytes per bit-plane row (include/breedhip.h, bh_packed_pitch): 16 per 64 markers, and whole 128-byte lines for
    rows beyond 4 KB."""
    nw = (n_markers + 63) // 64
    return 16 * (nw if nw <= 256 else (nw + 7) // 8 * 8)


class OracleEngine(EngineBase):
    def __init__(self, recombination_vec, marker_effects, device=None):
        self.device = torch.device("cpu")
        self.r = np.ascontiguousarray(recombination_vec, np.float32)
        eff = np.ascontiguousarray(marker_effects, np.float32)
        self.effects = eff if eff.ndim == 2 else eff[:, None]
        self.n_markers, self.n_traits = self.effects.shape
        self.pitch = (2 * self.n_markers + 15) // 16 * 16
        self.packed_pitch = packed_pitch_of(self.n_markers)
        self.calls = {"cross": 0, "gather": 0, "gebv": 0, "reward_max": 0, "topk": 0}
        self.threefry_layout = "original"
        self.env_window = (0, 0)
        self.cross_window = (0, 0)
        self.mutation = 0.0

    def set_mutation(self, probability):
        self.mutation = float(probability)

    def set_threefry_layout(self, layout):
        self.threefry_layout = layout

    def set_env_window(self, env_offset, envs_total):
        self.env_window = (int(env_offset), int(envs_total))

    def set_cross_window(self, cross_offset, crosses_total):
        self.cross_window = (int(cross_offset), int(crosses_total))

    def _cross_windowed(self, p, idx, key, rng_mode):
        """The oracle has no windows: cross the WHOLE batch the window speaks of (this shard's pairs in place,
        row 0 x row 0 elsewhere) and keep this shard's rows -- which is what the window means."""
        n_envs, n_out = idx.shape[:2]
        off, total = self.cross_window
        if not total and self.env_window[1]:
            off, total = self.env_window[0] * n_out, self.env_window[1] * n_out
        if not (rng_mode & 1) or not total:
            return co.cross(p, idx, self.r, key, rng_mode, mutation=self.mutation)
        assert n_envs * n_out + off <= total
        flat = np.zeros((1, total, 2), np.int32)
        # one flat batch: env e's pool rows are offset by e * pool_rows
        pool_rows = p.shape[1]
        flat[0, off : off + n_envs * n_out] = (idx + (np.arange(n_envs) * pool_rows)[:, None, None]).reshape(-1, 2)
        o, newkey = co.cross(p.reshape(1, n_envs * pool_rows, *p.shape[2:]), flat, self.r, key, rng_mode,
                              mutation=self.mutation)
        return o[0, off : off + n_envs * n_out].reshape(n_envs, n_out, *o.shape[2:]), newkey

    def _g(self, rows_t):  # (rows, pitch) -> (rows, M, 2) numpy u8
        return rows_t.numpy()[:, : 2 * self.n_markers].reshape(-1, self.n_markers, 2)

    def _upload_rows(self, a, validate):
        if validate and a.max(initial=0) > 1:
            raise ValueError("genotype bytes must be 0 or 1")
        out = torch.zeros((a.shape[0], self.pitch), dtype=torch.uint8)
        out[:, : 2 * self.n_markers] = torch.from_numpy(a.reshape(a.shape[0], -1))
        return out

    def download_population(self, rows_t):
        return self._g(rows_t).copy().view(np.bool_)

    def cross(self, pool, n_envs, pool_rows, idx, n_out, key, rng_mode, out, gebv=None, reward=None, reward_next=None):
        self.calls["cross"] += 1
        co.set_threefry_layout(self.threefry_layout)
        if reward_next is not None:
            reward_next[:n_envs] = float("-inf")
        p = self._g(pool)[: n_envs * pool_rows].reshape(n_envs, pool_rows, self.n_markers, 2)
        i = idx.numpy().reshape(n_envs, n_out, 2)
        i = np.clip(np.where(i < 0, i + pool_rows, i), 0, pool_rows - 1).astype(np.int32)  # JAX gather rules
        o, newkey = self._cross_windowed(p, i, key, rng_mode)
        key[:] = newkey
        out[: n_envs * n_out] = 0
        out[: n_envs * n_out, : 2 * self.n_markers] = torch.from_numpy(o.reshape(n_envs * n_out, -1))
        if gebv is not None:
            g = co.gebv(o.reshape(-1, self.n_markers, 2), self.effects)
            gebv.reshape(-1, self.n_traits)[: n_envs * n_out] = torch.from_numpy(g)
            if reward is not None:
                reward[:n_envs] = torch.from_numpy(co.reward_max(g.reshape(n_envs, -1)))

    def gather(self, src, idx, out, gebv=None):
        self.calls["gather"] += 1
        n = src.shape[0]
        i = idx.numpy().astype(np.int64).reshape(-1)
        i = np.clip(np.where(i < 0, i + n, i), 0, n - 1)
        out[: len(i)] = src[torch.from_numpy(i)]
        if gebv is not None:
            gebv.reshape(-1, self.n_traits)[: len(i)] = torch.from_numpy(co.gebv(self._g(out[: len(i)]), self.effects))

    # ---- bit-plane rows (layout documented in include/breedhip.h) ------------------------------
    def _pack_np(self, g):  # (rows, M, 2) u8 -> (rows, packed_pitch) u8: bit i of the row = allele byte i
        rows, M = g.shape[0], self.n_markers
        by = np.zeros((rows, self.packed_pitch * 8), np.uint8)
        by[:, : 2 * M] = g.reshape(rows, 2 * M) & 1
        return np.packbits(by, axis=1, bitorder="little")

    def _unpack_np(self, p):  # (rows, packed_pitch) u8 -> (rows, M, 2) u8
        bits = np.unpackbits(np.ascontiguousarray(p), axis=1, bitorder="little")
        return np.ascontiguousarray(bits[:, : 2 * self.n_markers].reshape(p.shape[0], self.n_markers, 2))

    def _put_bytes(self, out, g):
        out[: g.shape[0]] = 0
        out[: g.shape[0], : 2 * self.n_markers] = torch.from_numpy(g.reshape(g.shape[0], -1))

    def pack(self, rows_t, out=None):
        p = torch.from_numpy(self._pack_np(self._g(rows_t)))
        if out is not None:
            out[: p.shape[0]] = p
            return out
        return p

    def unpack(self, packed, out=None):
        if out is None:
            out = self.empty_population(packed.shape[0])
        self._put_bytes(out, self._unpack_np(packed.numpy()))
        return out

    def cross_packed(self, pool, n_envs, pool_rows, idx, n_out, key, rng_mode, out_packed, out_bytes=None, gebv=None,
                     reward=None, reward_next=None):
        tmp = self.empty_population(n_envs * n_out)
        self.cross(self.unpack(pool[: n_envs * pool_rows]), n_envs, pool_rows, idx, n_out, key, rng_mode, tmp, gebv, reward,
                   reward_next)
        out_packed[: n_envs * n_out] = self.pack(tmp)
        if out_bytes is not None:
            out_bytes[: n_envs * n_out] = tmp

    def gather_packed(self, src, idx, n_rows, out_packed=None, out_bytes=None, gebv=None, deferred=False):
        if idx is None:
            idx = torch.arange(n_rows, dtype=torch.int32)
        tmp = self.empty_population(n_rows)
        self.gather(self.unpack(src), idx, tmp, gebv)
        if out_packed is not None:
            out_packed[:n_rows] = self.pack(tmp)
        if out_bytes is not None:
            out_bytes[:n_rows] = tmp

    def gebv(self, pop, out=None):
        self.calls["gebv"] += 1
        g = torch.from_numpy(co.gebv(self._g(pop), self.effects))
        if out is not None:
            out.reshape(-1, self.n_traits)[:] = g
            return out
        return g

    def optimal_haploid(self, pop, block_start=None, want_pop=True, want_value=True):
        bs = None if block_start is None else block_start.numpy()
        ohp, ohv = co.optimal_haploid(self._g(pop), self.effects, bs)
        return (torch.from_numpy(ohp) if want_pop else None), (torch.from_numpy(ohv) if want_value else None)

    def reward_max(self, gebv, n_envs, out=None):
        self.calls["reward_max"] += 1
        r = torch.from_numpy(co.reward_max(gebv.numpy().reshape(n_envs, -1)))
        if out is not None:
            out[:] = r
            return out
        return r

    def topk(self, scores, k):
        self.calls["topk"] += 1
        s = scores.to(torch.float32).numpy()
        s2 = s.reshape(-1, s.shape[-1])
        out = np.stack([co.topk(row, k) for row in s2])
        return torch.from_numpy(out.reshape(*s.shape[:-1], k))

    def founders(self, env_key, n_envs, n_germplasm, n, deferred=False):
        co.set_threefry_layout(self.threefry_layout)
        off, total = self.env_window
        if total:  # the draw of the whole, this shard's envs kept
            f, k = co.founders(env_key, total, n_germplasm, n)
            return torch.from_numpy(f[off : off + n_envs].reshape(-1).copy()), k
        f, k = co.founders(env_key, n_envs, n_germplasm, n)
        return torch.from_numpy(f.reshape(-1)), k

    def selection_actions(self, scores, k, n_crosses, env_key):
        s = scores.as_subclass(torch.Tensor).to(torch.float32).numpy() if isinstance(scores, torch.Tensor) else np.asarray(scores, np.float32)
        co.set_threefry_layout(self.threefry_layout)
        off, total = self.env_window
        if total:
            whole = np.zeros((total, s.shape[1]), np.float32)
            whole[off : off + s.shape[0]] = s
            a, key = co.selection_actions(whole, k, n_crosses, env_key)
            return torch.from_numpy(a[off : off + s.shape[0]].copy()), key
        a, key = co.selection_actions(s, k, n_crosses, env_key)
        return torch.from_numpy(a), key

    def winner_layout(self, winners, n_local, world, rank, cap, pairs=None):
        from oracle import numpy_oracle as no

        send, pos, counts, out = no.winner_layout(winners.numpy(), n_local, world, rank, cap,
                                                  None if pairs is None else pairs.numpy())
        return (torch.from_numpy(send), torch.from_numpy(pos), torch.from_numpy(counts),
                None if out is None else torch.from_numpy(out.reshape(pairs.shape)))

    def sync(self):
        pass

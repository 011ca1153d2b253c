"""The device engine: genotype rows in HBM and the HIP operators that act on them.

``HipEngine`` is a thin Python face of one ``bh_ctx`` (include/breedhip.h).  torch is used for
what it is good at here -- owning device memory and the stream -- and nothing else: every
operator is a hand-written HIP kernel behind the C ABI, called with raw device pointers.

Layout: a population is a ``(rows, pitch)`` uint8 tensor, ``pitch = 2*M`` rounded up to 16
bytes; bytes ``[2m, 2m+1]`` of a row are the two alleles of marker ``m`` (the reference's
``(N, M, 2)`` bool layout, one byte per allele).  ``as_genotypes`` gives the zero-copy
``(rows, M, 2)`` bool view.
"""
import ctypes
import math
import sys
from typing import Optional

import numpy as np
import torch

from . import _lib


class DeviceArray(torch.Tensor):
    """A torch.Tensor that converts to NumPy implicitly, the way a ``jax.Array`` does.

    The reference hands JAX arrays to user code that feeds them straight to NumPy
    (``np.max(infos["GEBV"], axis=(1, 2))``, vec_env.py:97; ``np.add.outer`` in
    tests/test_vec.py:198).  A plain CUDA tensor refuses that; this subclass copies to host
    on ``__array__`` instead.  Everything else is ordinary torch.
    """

    @staticmethod
    def wrap(t: torch.Tensor) -> "DeviceArray":
        return t.as_subclass(DeviceArray)

    def __array__(self, dtype=None, copy=None):
        a = self.detach().as_subclass(torch.Tensor).cpu().numpy()
        return a.astype(dtype) if dtype is not None else a

    def block_until_ready(self):
        if self.is_cuda:
            torch.cuda.current_stream(self.device).synchronize()
        return self


def _numpy_style(name):
    """``np.max(x, axis=(1, 2))`` calls ``x.max(axis=..., out=None)``: answer NumPy-style calls in
    NumPy (on the host copy), leave torch-style calls (``dim=``, positional) to torch."""
    torch_fn = getattr(torch.Tensor, name)

    def method(self, *args, **kwargs):
        if "axis" in kwargs or "out" in kwargs or "keepdims" in kwargs:
            kwargs.pop("out", None)
            return getattr(np.asarray(self), name)(*args, **kwargs)
        return torch_fn(self, *args, **kwargs)

    method.__name__ = name
    return method


for _name in ("max", "min", "sum", "mean", "std", "var", "argmax", "argmin", "any", "all", "prod"):
    setattr(DeviceArray, _name, _numpy_style(_name))


class PackedPopulation:
    """A population kept as bit-plane rows: what ``VecBreedGym(storage="bits", observations="lazy")``
    hands out in place of the byte observation.  It has the observation's ``shape``; ``unpack()``
    (or ``np.asarray``) expands it to the usual ``(*lead, M, 2)`` bool genotypes on demand."""

    dtype = np.dtype(np.bool_)

    def __init__(self, engine, packed_rows: torch.Tensor, lead):
        self.engine, self.packed_rows, self.lead = engine, packed_rows, tuple(lead)
        self._bytes = None

    @property
    def shape(self):
        return (*self.lead, self.engine.n_markers, 2)

    def __len__(self):
        return self.lead[0]

    def unpack(self) -> DeviceArray:
        if self._bytes is None:
            self._bytes = self.engine.as_genotypes(self.engine.unpack(self.packed_rows), *self.lead)
        return self._bytes

    def __array__(self, dtype=None, copy=None):
        return self.unpack().__array__(dtype)

    def __getitem__(self, item):
        return self.unpack()[item]


# How many owners a storage has, Python wrappers and C++ holders (a view, a DLPack capsule) alike.  A private symbol of
# torch: looked up by name, and CHECKED (block_pool_supported) before the pool relies on it.
_storage_use_count = getattr(torch._C, "_storage_Use_Count", None)


class _Blocks:
    """A set of device blocks with the views handed out of them, and a cheap test for "nobody outside still holds
    any of it": the Python reference counts of the handed-out objects and the use counts of their storages (a slice
    or a ``view`` a caller made shares the storage) are what they were when only the pool held them.

    What this sees: the objects themselves, views / slices / ``detach()`` of them, DLPack exports
    (``torch.utils.dlpack.to_dlpack``, ``from_dlpack`` consumers), anything that keeps the tensor object alive
    (NumPy arrays of a CPU tensor, CuPy / Numba arrays made through ``__cuda_array_interface__``, which hold the
    exporting object).  What it cannot see: a raw ``data_ptr()`` kept elsewhere, the bare ``__cuda_array_interface__``
    dict without its tensor, and work a consumer enqueued on ANOTHER stream before dropping its reference (torch's
    caching allocator has the same rule: keep the tensor, or ``record_stream``)."""

    def __init__(self, tensors, handed_out):
        self.storages = tuple(t.untyped_storage() for t in tensors)  # kept: their use counts include these wrappers
        self.cdata = tuple(st._cdata for st in self.storages)
        self.handed_out = tuple(handed_out)
        self.base = None  # rebase(): once the pool alone holds the objects

    def rebase(self):
        self.base = self._counts()

    def _counts(self, refcount=sys.getrefcount):
        use_count = _storage_use_count
        return [use_count(c) for c in self.cdata] + [refcount(o) for o in self.handed_out]

    def free(self) -> bool:
        return self._counts() == self.base


_pool_ok = None


def block_pool_supported() -> bool:
    """Whether this torch build lets ``_Blocks.free`` see every kind of holder it promises to see: checked once,
    on CPU tensors (storages are counted the same way on every device), before the first pool is made.  A build
    where the private use-count symbol is gone, or where a C++ owner no longer shows in either count, gets fresh
    blocks per step (``block_pool=0`` behaviour) and a warning -- never a silently overwritten observation."""
    global _pool_ok
    if _pool_ok is not None:
        return _pool_ok
    ok = _storage_use_count is not None
    if ok:
        try:
            t = torch.empty(64, dtype=torch.uint8)
            b = _Blocks([t], [t])
            del t  # as in StepRing._add_slot: the counts of "nobody else holds it" are taken without local names
            b.rebase()
            probes = (lambda x: x[3:9], lambda x: x.view(torch.int8), lambda x: x.detach(), lambda x: [x],
                      torch.utils.dlpack.to_dlpack, lambda x: torch.from_dlpack(x), lambda x: x.numpy())
            for hold in probes:
                h = hold(b.handed_out[0])
                seen = not b.free()
                del h
                ok = ok and seen and b.free()
        except Exception:
            ok = False
    _pool_ok = ok
    return ok


class StepRing:
    """The output blocks of ``VecBreedGym.step``, kept and reused instead of allocated per step (include/breedhip.h,
    bh_step_bind): a generation slot holds the bit-plane state, the byte-layout observation (unless the env hands out
    lazy ones) and the GEBV, each with the Python views the env returns already made; rewards have small blocks of
    their own (a launch primes the block of the NEXT step's reward).  ``step`` is the whole host side of a
    generation: find a free slot, one ``bh_step`` call.

    A slot is reused only when nothing handed out of it is referenced any more (``_Blocks.free``), so what a caller
    keeps stays valid for as long as it is kept -- the reference's arrays behave that way -- and a loop that drops
    what it got alternates between two slots: the 164 MB of two observation blocks of 16 x 256 x 10k stay in the
    256 MB Infinity Cache, which a fixed ring of three or more does not (24.5 against 26.3-29.3 us per step,
    profiles/r03_ring_depth.txt).  When every slot is held the pool grows, up to ``max_slots``."""

    OBS = 4  # bh_step flag

    def __init__(self, eng, n_envs, n_out, with_observations, max_slots=16):
        self.eng, self.n_envs, self.n_out = eng, int(n_envs), int(n_out)
        self.with_observations, self.max_slots = bool(with_observations), int(max_slots)
        self.slots, self.state, self.obs_rows, self.populations, self.gebv, self.lazy = [], [], [], [], [], []
        self.rewards, self.reward = [], []
        self.reserved = None  # the reward block the last launch primed for the next step
        self._bh_step, self._h = eng._L.bh_step, eng._h
        self._key_obj = self._key_ptr = None
        for _ in range(2):
            self._add_slot()
        for _ in range(4):
            self._add_reward()
        self._bind()

    def _add_slot(self):
        eng, rows, dev = self.eng, self.n_envs * self.n_out, self.eng.device
        state = torch.empty((rows, eng.packed_pitch), dtype=torch.uint8, device=dev)
        obs = torch.empty((rows, eng.pitch), dtype=torch.uint8, device=dev) if self.with_observations else None
        gebv = torch.empty((self.n_envs, self.n_out, eng.n_traits), dtype=torch.float32, device=dev).as_subclass(DeviceArray)
        pops = None if obs is None else eng.as_genotypes(obs, self.n_envs, self.n_out)
        lazy = PackedPopulation(eng, state, (self.n_envs, self.n_out))
        self.state.append(state)
        self.obs_rows.append(obs)
        self.populations.append(pops)
        self.gebv.append(gebv)
        self.lazy.append(lazy)
        self.slots.append(_Blocks([t for t in (state, obs, gebv) if t is not None],
                                  [o for o in (state, obs, pops, gebv, lazy) if o is not None]))
        del state, obs, gebv, pops, lazy  # the counts of "nobody else holds it" are taken without these names
        self.slots[-1].rebase()

    def _add_reward(self):
        r = torch.empty_strided((self.n_envs,), (64,), dtype=torch.float32, device=self.eng.device).as_subclass(DeviceArray)
        self.reward.append(r)
        self.rewards.append(_Blocks([r], [r]))
        del r
        self.rewards[-1].rebase()

    def _bind(self):
        def table(ts):
            return (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])

        _lib.check(self.eng._L.bh_step_bind(
            self._h, self.n_envs, self.n_out, len(self.state), table(self.state),
            table(self.obs_rows) if self.with_observations else None, table(self.gebv), len(self.reward),
            table(self.reward), 64))

    def take(self):
        """A generation slot nobody holds any more (the pool grows when there is none; None when it may not)."""
        for s, blocks in enumerate(self.slots):
            if blocks.free():
                self.lazy[s]._bytes = None
                return s
        if len(self.slots) >= self.max_slots:
            return None
        self._add_slot()
        self._bind()
        return len(self.slots) - 1

    def take_reward(self, but=None):
        for r, blocks in enumerate(self.rewards):
            if r != but and r != self.reserved and blocks.free():
                return r
        if len(self.rewards) >= 4 * self.max_slots:
            return None
        self._add_reward()
        self._bind()
        return len(self.rewards) - 1

    def step(self, pool_ptr: int, pool_rows: int, idx: torch.Tensor, key: np.ndarray, flags: int, want_reward: bool,
             want_next: bool):
        """One generation into a free slot -> (slot, reward block or None); None when no slot may be taken.
        ``want_next``: prime a reward block for the next step that has a reward."""
        s = self.take()
        if s is None:
            return None
        r = -1
        if want_reward:
            r, self.reserved = self.reserved, None
            if r is None:
                r = self.take_reward()  # not primed: the library fills it first
                if r is None:
                    return None
        nxt = -1
        if want_next:
            nxt = self.take_reward(but=r)
            self.reserved = nxt
            if nxt is None:
                nxt = -1
        if key is not self._key_obj:
            assert key.dtype == np.uint32 and key.shape == (2,) and key.flags.c_contiguous
            self._key_obj, self._key_ptr = key, key.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))
        rc = self._bh_step(self._h, s, r, nxt, pool_ptr, pool_rows, idx.data_ptr(), self._key_ptr, flags)
        if rc:
            _lib.check(rc)
        return s, (r if r >= 0 else None)

    def close(self):
        if self.eng._h:
            self.eng._L.bh_step_bind(self.eng._h, 0, 0, 0, None, None, None, 0, None, 1)


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _reward_pitch(reward, reward_next) -> int:
    """Floats between two envs' words in the reward buffers of a cross call (both must agree)."""
    pitches = {int(t.stride(0)) if t.numel() > 1 else None for t in (reward, reward_next) if t is not None}
    pitches.discard(None)
    if len(pitches) > 1:
        raise ValueError("reward and reward_next must have the same stride")
    return pitches.pop() if pitches else 1


class EngineBase:
    """Row layout and memory helpers shared by every engine; operators are abstract.

    Attributes an engine provides: ``device``, ``n_markers``, ``n_traits``, ``pitch``."""

    # ---- memory ------------------------------------------------------------------------
    def empty_population(self, rows: int) -> torch.Tensor:
        return torch.empty((int(rows), self.pitch), dtype=torch.uint8, device=self.device)

    def empty(self, shape, dtype) -> torch.Tensor:
        return torch.empty(shape, dtype=dtype, device=self.device)

    def empty_reward(self, n_envs: int) -> torch.Tensor:
        """(n_envs,) f32 reward buffer, every env's word on a cache line of its own (the atomic maxima of a launch
        then do not queue behind each other): a strided view."""
        return torch.empty_strided((int(n_envs),), (64,), dtype=torch.float32, device=self.device)

    def empty_packed(self, rows: int) -> torch.Tensor:
        """Bit-plane rows: ``packed_pitch`` bytes each -- 16 per 64 markers, whole 128-byte lines beyond 4 KB
        (``bh_packed_pitch``, include/breedhip.h)."""
        return torch.empty((int(rows), self.packed_pitch), dtype=torch.uint8, device=self.device)

    def as_genotypes(self, rows_t: torch.Tensor, *lead) -> DeviceArray:
        """(rows, pitch) uint8 -> zero-copy (*lead, M, 2) bool view."""
        rows = rows_t.shape[0]
        if not lead:
            lead = (rows,)
        if self.pitch == 2 * self.n_markers and rows_t.is_contiguous() and rows == math.prod(lead):
            # no pad bytes: a plain reshape
            return rows_t.view(torch.bool).view(*lead, self.n_markers, 2).as_subclass(DeviceArray)
        g = rows_t[:, : 2 * self.n_markers].view(torch.bool)
        strides, acc = [], self.pitch
        for d in reversed(lead):
            strides.append(acc)
            acc *= d
        strides = tuple(reversed(strides))
        return DeviceArray.wrap(g.as_strided((*lead, self.n_markers, 2), (*strides, 2, 1)))

    def upload_population(self, pop, validate: bool = True) -> torch.Tensor:
        """host (rows, M, 2) bool/uint8 (or an already resident (rows, pitch) tensor) -> device rows."""
        if isinstance(pop, torch.Tensor):
            pop = pop.as_subclass(torch.Tensor)
            if pop.device == self.device:
                if pop.dtype == torch.uint8 and pop.dim() == 2 and pop.shape[1] == self.pitch:
                    return pop
                return self.pack_device(pop)
            pop = pop.cpu().numpy()
        a = np.asarray(pop)
        if a.ndim != 3 or a.shape[1] != self.n_markers or a.shape[2] != 2:
            raise ValueError(f"population must have shape (n, {self.n_markers}, 2), got {a.shape}")
        a = np.ascontiguousarray(a.view(np.uint8) if a.dtype == np.bool_ else a.astype(np.uint8))
        return self._upload_rows(a, validate)

    def _upload_rows(self, a: np.ndarray, validate: bool) -> torch.Tensor:
        raise NotImplementedError

    def pack_device(self, g: torch.Tensor) -> torch.Tensor:
        """device (rows, M, 2) bool/uint8 tensor -> (rows, pitch) rows (a device-side copy)."""
        g = g.reshape(-1, 2 * self.n_markers)
        g = g.view(torch.uint8) if g.dtype == torch.bool else g.to(torch.uint8)
        if self.pitch == 2 * self.n_markers:
            return g.contiguous()
        out = torch.zeros((g.shape[0], self.pitch), dtype=torch.uint8, device=self.device)
        out[:, : 2 * self.n_markers] = g
        return out

    def download_population(self, rows_t: torch.Tensor) -> np.ndarray:
        raise NotImplementedError

    def to_device(self, a, dtype) -> torch.Tensor:
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=dtype).contiguous()
        return torch.as_tensor(np.ascontiguousarray(a)).to(device=self.device, dtype=dtype)


class HipEngine(EngineBase):
    """One GPU, one genetic map: thresholds and marker effects resident in HBM."""

    def __init__(self, recombination_vec: np.ndarray, marker_effects: np.ndarray, device=None,
                 threefry_layout="original"):
        L = _lib.lib()
        if not torch.cuda.is_available() or _lib.device_count() == 0:
            raise RuntimeError(
                "breedgym_amd needs an AMD GPU (gfx950): no HIP device is visible and there is no CPU fallback"
            )
        if device is None:
            device = torch.cuda.current_device()
        self.device = torch.device("cuda", device if isinstance(device, int) else torch.device(device).index or 0)
        r = np.ascontiguousarray(recombination_vec, np.float32)
        eff = np.ascontiguousarray(marker_effects, np.float32)
        if eff.ndim == 1:
            eff = eff[:, None]
        if r.ndim != 1 or eff.shape[0] != r.shape[0]:
            raise ValueError("recombination_vec must be (M,) and marker_effects (M, T)")
        self.n_markers, self.n_traits = int(eff.shape[0]), int(eff.shape[1])
        h = ctypes.c_void_p()
        _lib.check(
            L.bh_create(
                self.device.index, self.n_markers, self.n_traits,
                r.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                eff.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), ctypes.byref(h),
            )
        )
        self._h = h
        self._L = L
        self.pitch = int(L.bh_row_pitch(h))
        self.packed_pitch = int(L.bh_packed_pitch(h))
        self.threefry_layout = "original"
        self.set_threefry_layout(threefry_layout)
        self._key_obj, self._key_ptr = None, None
        # torch's default stream is the null stream, which is also the context's default

    def close(self):
        if getattr(self, "_h", None):
            self._L.bh_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- memory ------------------------------------------------------------------------
    def _upload_rows(self, a: np.ndarray, validate: bool) -> torch.Tensor:
        out = self.empty_population(a.shape[0])
        _lib.check(self._L.bh_upload_population(self._h, a.ctypes.data, a.shape[0], out.data_ptr(), int(validate)))
        return out

    def download_population(self, rows_t: torch.Tensor) -> np.ndarray:
        rows = rows_t.shape[0]
        out = np.empty((rows, self.n_markers, 2), np.uint8)
        _lib.check(self._L.bh_download_population(self._h, rows_t.data_ptr(), rows, out.ctypes.data))
        return out.view(np.bool_)

    # ---- operators ---------------------------------------------------------------------
    def cross(self, pool, n_envs, pool_rows, idx, n_out, key, rng_mode, out, gebv=None, reward=None,
              reward_next=None) -> None:
        """Fused gather + meiosis + GEBV (+ per-env max reward) (bh_cross).  ``key`` (uint32[2]) is advanced in place.
        ``reward_next``: (n_envs,) f32 the launch leaves holding -inf, to be the ``reward`` of the next call."""
        assert key.dtype == np.uint32 and key.shape == (2,) and key.flags.c_contiguous
        assert idx.dtype == torch.int32 and idx.is_contiguous() and idx.numel() == n_envs * n_out * 2
        assert pool.shape[0] >= n_envs * pool_rows and out.shape[0] >= n_envs * n_out
        assert gebv is None or (gebv.dtype == torch.float32 and gebv.numel() >= n_envs * n_out * self.n_traits)
        _lib.check(
            self._L.bh_cross(
                self._h, pool.data_ptr(), n_envs, pool_rows, idx.data_ptr(), n_out,
                key.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), int(rng_mode), out.data_ptr(), _ptr(gebv),
                _ptr(reward), _ptr(reward_next), _reward_pitch(reward, reward_next),
            )
        )

    def gather(self, src, idx, out, gebv=None) -> None:
        assert idx.dtype == torch.int32 and idx.is_contiguous() and out.shape[0] >= idx.numel()
        _lib.check(
            self._L.bh_gather(self._h, src.data_ptr(), src.shape[0], idx.data_ptr(), idx.numel(), out.data_ptr(), _ptr(gebv))
        )

    # ---- bit-plane rows (1 bit per allele; include/breedhip.h, "bit-plane storage") -------------
    def pack(self, rows_t, out=None) -> torch.Tensor:
        """(rows, pitch) byte-layout rows -> (rows, packed_pitch) packed rows."""
        if out is None:
            out = self.empty_packed(rows_t.shape[0])
        _lib.check(self._L.bh_pack(self._h, rows_t.data_ptr(), rows_t.shape[0], out.data_ptr()))
        return out

    def unpack(self, packed, out=None) -> torch.Tensor:
        """(rows, packed_pitch) packed rows -> (rows, pitch) byte-layout rows."""
        if out is None:
            out = self.empty_population(packed.shape[0])
        self.gather_packed(packed, None, packed.shape[0], None, out)
        return out

    def cross_packed(self, pool, n_envs, pool_rows, idx, n_out, key, rng_mode, out_packed, out_bytes=None,
                     gebv=None, reward=None, reward_next=None) -> None:
        """``cross`` on a packed pool (bh_cross_packed); ``out_bytes`` also receives the byte-layout offspring."""
        assert key.dtype == np.uint32 and key.shape == (2,) and key.flags.c_contiguous
        assert idx.dtype == torch.int32 and idx.is_contiguous() and idx.numel() == n_envs * n_out * 2
        assert pool.shape[0] >= n_envs * pool_rows and out_packed.shape[0] >= n_envs * n_out
        assert out_bytes is None or out_bytes.shape[0] >= n_envs * n_out
        assert gebv is None or (gebv.dtype == torch.float32 and gebv.numel() >= n_envs * n_out * self.n_traits)
        _lib.check(
            self._L.bh_cross_packed(
                self._h, pool.data_ptr(), n_envs, pool_rows, idx.data_ptr(), n_out,
                key.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), int(rng_mode), out_packed.data_ptr(),
                _ptr(out_bytes), _ptr(gebv), _ptr(reward), _ptr(reward_next), _reward_pitch(reward, reward_next),
            )
        )

    def cross_step(self, pool, n_envs, pool_rows, idx, n_out, key, rng_mode, want_obs, reward, want_next):
        """``VecBreedGym.step``'s call, thin: allocates this generation's blocks and makes the one C call
        (bh_cross_packed), nothing else -- at 16 x 256 x 10k the launch takes 23 us and the host path must stay
        below it.  ``idx`` must be an int32 device tensor (E, n_out, 2), ``key`` a C-contiguous uint32[2] array
        (advanced in place), ``reward`` None or a buffer from an earlier call's ``want_next``.
        -> (packed rows, byte rows or None, gebv (E, n_out, T), the next step's primed reward buffer or None)"""
        dev, rows = self.device, n_envs * n_out
        outp = torch.empty((rows, self.packed_pitch), dtype=torch.uint8, device=dev)
        out = torch.empty((rows, self.pitch), dtype=torch.uint8, device=dev) if want_obs else None
        gebv = torch.empty((n_envs, n_out, self.n_traits), dtype=torch.float32, device=dev)
        nxt = torch.empty_strided((n_envs,), (64,), dtype=torch.float32, device=dev) if want_next else None
        if key is not self._key_obj:  # the simulator keeps one key array: its address is looked up once
            assert key.dtype == np.uint32 and key.shape == (2,) and key.flags.c_contiguous
            self._key_obj, self._key_ptr = key, key.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))
        rc = self._L.bh_cross_packed(
            self._h, pool.data_ptr(), n_envs, pool_rows, idx.data_ptr(), n_out, self._key_ptr, rng_mode,
            outp.data_ptr(), None if out is None else out.data_ptr(), gebv.data_ptr(),
            None if reward is None else reward.data_ptr(), None if nxt is None else nxt.data_ptr(), 64,
        )
        if rc:
            _lib.check(rc)
        return outp, out, gebv, nxt

    def step_ring(self, n_envs: int, n_out: int, with_observations: bool, max_slots: int = 16) -> Optional["StepRing"]:
        """Reusable output blocks for ``VecBreedGym.step``, bound to the context (bh_step_bind): a step then is one
        C call with no allocation (``StepRing.step``).  None (= fresh blocks per step) when this torch build fails
        ``block_pool_supported``."""
        if not block_pool_supported():
            import warnings

            warnings.warn(
                f"breedgym_amd: torch {torch.__version__} does not let the block pool see who still holds a block "
                "(torch._C._storage_Use_Count is missing or no longer counts C++ owners): block_pool is off, every "
                "step allocates fresh blocks", RuntimeWarning, stacklevel=2)
            return None
        return StepRing(self, n_envs, n_out, with_observations, max_slots)

    def gather_packed(self, src, idx, n_rows, out_packed=None, out_bytes=None, gebv=None, deferred=False) -> None:
        """Rows ``idx`` (or 0..n_rows-1) of packed ``src`` -> packed and/or byte-layout rows and/or GEBV.
        ``deferred``: asked for ahead of a ``cross_packed`` / ``StepRing.step`` call, whose launch then makes the gather
        as well (bh_gather_packed_deferred); the caller keeps every tensor alive until that call."""
        assert idx is None or (idx.dtype == torch.int32 and idx.is_contiguous() and idx.numel() == n_rows)
        assert out_packed is None or out_packed.shape[0] >= n_rows
        assert out_bytes is None or out_bytes.shape[0] >= n_rows
        _lib.check(
            (self._L.bh_gather_packed_deferred if deferred else self._L.bh_gather_packed)(
                self._h, src.data_ptr(), src.shape[0], _ptr(idx), int(n_rows), _ptr(out_packed), _ptr(out_bytes),
                _ptr(gebv),
            )
        )

    def gebv(self, pop, out=None) -> torch.Tensor:
        rows = pop.shape[0]
        if out is None:
            out = torch.empty((rows, self.n_traits), dtype=torch.float32, device=self.device)
        _lib.check(self._L.bh_gebv(self._h, pop.data_ptr(), rows, out.data_ptr()))
        return out

    def optimal_haploid(self, pop, block_start=None, want_pop=True, want_value=True):
        """Optimal haploid (rows, M) uint8 and its value (rows, T) f32 of byte-layout rows (bh_optimal_haploid);
        ``block_start``: int32 device tensor of block offsets, or None for the blockless rule."""
        rows = pop.shape[0]
        ohp = torch.empty((rows, self.n_markers), dtype=torch.uint8, device=self.device) if want_pop else None
        ohv = torch.empty((rows, self.n_traits), dtype=torch.float32, device=self.device) if want_value else None
        nb = 0 if block_start is None else block_start.numel() - 1
        _lib.check(self._L.bh_optimal_haploid(self._h, pop.data_ptr(), rows, _ptr(block_start), nb, _ptr(ohp), _ptr(ohv)))
        return ohp, ohv

    def reward_max(self, gebv, n_envs, out=None) -> torch.Tensor:
        if out is None:
            out = torch.empty((n_envs,), dtype=torch.float32, device=self.device)
        _lib.check(self._L.bh_reward_max(self._h, gebv.data_ptr(), n_envs, gebv.numel() // max(n_envs, 1), out.data_ptr()))
        return out

    def topk(self, scores, k) -> torch.Tensor:
        s = scores.to(device=self.device, dtype=torch.float32).contiguous()
        s2 = s.reshape(-1, s.shape[-1])
        out = torch.empty((s2.shape[0], int(k)), dtype=torch.int32, device=self.device)
        _lib.check(self._L.bh_topk(self._h, s2.data_ptr(), s2.shape[0], s2.shape[1], int(k), out.data_ptr()))
        return out.reshape(*s.shape[:-1], int(k))

    def founders(self, env_key: np.ndarray, n_envs: int, n_germplasm: int, n: int, deferred: bool = False):
        """Founder draw of ``VecBreedGym.reset`` on the device -> ((n_envs * n,) int32 tensor, new key).
        ``deferred``: asked for ahead of a ``cross_packed`` call, whose launch then makes the draw as well."""
        k = np.ascontiguousarray(env_key, np.uint32).copy()
        out = torch.empty((int(n_envs) * int(n),), dtype=torch.int32, device=self.device)
        _lib.check(
            (self._L.bh_founders_deferred if deferred else self._L.bh_founders_device)(
                self._h, k.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), int(n_envs), int(n_germplasm), int(n),
                out.data_ptr(),
            )
        )
        return out, k

    def selection_actions(self, scores, k: int, n_crosses: int, env_key: np.ndarray):
        """``SelectionScores._convert_actions`` for every env in one launch -> ((E, n, 2) int32, new key)."""
        s = scores.as_subclass(torch.Tensor) if isinstance(scores, torch.Tensor) else torch.as_tensor(np.asarray(scores))
        s = s.to(device=self.device, dtype=torch.float32).contiguous()
        E, n = s.shape
        key = np.ascontiguousarray(env_key, np.uint32).copy()
        out = torch.empty((E, n, 2), dtype=torch.int32, device=self.device)
        _lib.check(
            self._L.bh_selection_actions(
                self._h, s.data_ptr(), E, n, int(k), int(n_crosses),
                key.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), out.data_ptr(),
            )
        )
        return out, key

    def pair_actions(self, scores, n: int) -> torch.Tensor:
        """``PairScores._convert_actions`` for every env (bh_pair_actions): (E, ...) pair scores -> (E, n, 2) int32.
        An env's scores are taken flat, whatever their shape (``action.flatten()``, vec_wrappers.py:102): (n, n) as
        documented, (n, T, E, n, T) when a caller feeds ``np.add.outer`` of multi-trait GEBVs (tests/test_vec.py:198)."""
        s = scores.as_subclass(torch.Tensor) if isinstance(scores, torch.Tensor) else torch.as_tensor(np.asarray(scores))
        s = s.to(device=self.device, dtype=torch.float32).contiguous()
        E = s.shape[0]
        out = torch.empty((E, n, 2), dtype=torch.int32, device=self.device)
        _lib.check(self._L.bh_pair_actions(self._h, s.data_ptr(), E, s.numel() // max(E, 1), int(n), out.data_ptr()))
        return out

    def winner_layout(self, winners, n_local: int, world: int, rank: int, cap: int, pairs=None):
        """Population sharding (bh_winner_layout): from the replicated list of winners (k,) int32 -> (send_idx (cap,),
        pos (k,), counts (world + 1,) [last: the largest], pairs remapped to rows of the gathered buffer or None), all
        on the device, nothing synchronised."""
        assert winners.dtype == torch.int32 and winners.is_contiguous()
        k = winners.numel()
        send = torch.empty((int(cap),), dtype=torch.int32, device=self.device)
        pos = torch.empty((k,), dtype=torch.int32, device=self.device)
        counts = torch.empty((int(world) + 1,), dtype=torch.int32, device=self.device)
        out = None
        if pairs is not None:
            assert pairs.dtype == torch.int32 and pairs.is_contiguous() and pairs.device == self.device
            out = torch.empty_like(pairs)
        _lib.check(self._L.bh_winner_layout(
            self._h, winners.data_ptr(), k, int(n_local), int(world), int(rank), int(cap), send.data_ptr(), pos.data_ptr(),
            counts.data_ptr(), _ptr(pairs), 0 if pairs is None else pairs.numel(), _ptr(out)))
        return send, pos, counts, out

    def set_threefry_layout(self, layout) -> None:
        """"original" (jax_threefry_partitionable off: what the reference's seeded tests were recorded under) or
        "partitionable" (the default of jax >= 0.5): which JAX bit stream the device paths reproduce."""
        code = _lib.layout_code(layout)
        _lib.check(self._L.bh_set_threefry_layout(self._h, code))
        self.threefry_layout = "partitionable" if code else "original"

    def set_mutation(self, probability: float) -> None:
        """chromax's ``Simulator(mutation=p)``: every gamete allele flips with probability p (0: off)."""
        _lib.check(self._L.bh_set_mutation(self._h, float(probability)))
        self.mutation = float(probability)

    def set_env_window(self, env_offset: int, envs_total: int) -> None:
        """This engine's envs are envs [env_offset, ..) of envs_total: founder draws, selection shuffles and
        independent crossover keys are those of the whole (include/breedhip.h, "sharding windows")."""
        _lib.check(self._L.bh_set_env_window(self._h, int(env_offset), int(envs_total)))

    def set_cross_window(self, cross_offset: int, crosses_total: int) -> None:
        """One batch of crosses cut row-wise over several engines: cross b here is cross cross_offset + b."""
        _lib.check(self._L.bh_set_cross_window(self._h, int(cross_offset), int(crosses_total)))

    def hint_next_cross(self, n_envs: int, n_out: int, rng_mode: int) -> None:
        """Say what the cross call AFTER the next one will be (bh_hint_next_cross): the next call's launch then makes
        that call's crossover masks, not a repeat of its own."""
        _lib.check(self._L.bh_hint_next_cross(self._h, int(n_envs), int(n_out), int(rng_mode)))

    def set_mask_prefetch(self, enable: bool) -> None:
        """Run-ahead of the next generation's crossover masks on a side stream (default on)."""
        _lib.check(self._L.bh_set_mask_prefetch(self._h, int(bool(enable))))

    def set_stream(self, stream=None) -> None:
        """Enqueue this engine's launches on ``stream`` (a ``torch.cuda.Stream``; None = the default stream).
        Tensors the calls use must belong to that stream as far as torch's caching allocator is concerned: make the
        calls while it is torch's current stream (``on_stream`` does both) or ``record_stream`` them."""
        handle = 0 if stream is None else int(stream.cuda_stream)
        _lib.check(self._L.bh_set_stream(self._h, ctypes.c_void_p(handle or None)))
        self._stream = stream

    def on_stream(self, stream):
        """Context manager: the engine's launches AND torch's allocations go to ``stream`` inside the block."""
        import contextlib

        @contextlib.contextmanager
        def scope():
            prev = getattr(self, "_stream", None)
            self.set_stream(stream)
            try:
                with torch.cuda.stream(stream):
                    yield self
            finally:
                self.set_stream(prev)

        return scope()

    def sync(self) -> None:
        _lib.check(self._L.bh_sync(self._h))

    def clear_error(self) -> int:
        """Recover the context after a device-side error (bh_clear_error; include/breedhip.h says what is sticky and
        what the recovery restores) -> the error word that was set (0: none).  Results of the calls between the event
        and this one stay untrusted."""
        was = ctypes.c_uint32(0)
        _lib.check(self._L.bh_clear_error(self._h, ctypes.byref(was)))
        return int(was.value)

    def release_scratch(self) -> None:
        """Give back the scratch the context has grown for its larger calls (bh_release_scratch)."""
        _lib.check(self._L.bh_release_scratch(self._h))

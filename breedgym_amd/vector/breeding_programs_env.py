"""``WheatBreedGym``: the wheat breeding scheme as a vector wrapper
(reference ``breedgym/vector/breeding_programs_env.py:12-72``).

Per step: the ``n_lines`` best-scoring pairs are crossed (softmax-proportional, as in
``PairScores``) -> every F1 gives ``plant_per_line`` doubled-haploid lines -> the ``k_per_line``
best plants of each line are kept -> the ``individual_per_gen`` best of those form the next
population.  Everything runs on the device through the same kernels as the plain env: the fused
cross launch, its double-haploid key mode, batched top-k and row gathers.

[EXTERNAL/UNVERIFIED] ``double_haploid`` / ``select`` are chromax members whose source is not in the
reference tree; the per-line selection follows the intent shown in ``scripts/time_wheat.py:20-24``
(``vmap(simulator.select)`` over lines).
"""
import torch

from .. import _lib, spaces
from ..engine import DeviceArray
from .vec_wrappers import PairScores, VectorWrapper


class WheatBreedGym(VectorWrapper):
    def __init__(self, vec_env, n_lines=200, plant_per_line=100, k_per_line=5):
        super().__init__(vec_env)
        self.n_lines = n_lines
        self.plant_per_line = plant_per_line
        self.k_per_line = k_per_line
        if k_per_line > plant_per_line:
            raise ValueError("k_per_line cannot exceed plant_per_line")
        if self.individual_per_gen > n_lines * k_per_line:
            raise ValueError("individual_per_gen cannot exceed n_lines * k_per_line")
        action_shape = self.n_lines, self.n_lines
        self.single_action_space = spaces.Box(-1e5, 1e5, shape=action_shape)
        self.action_space = spaces.Box(-1e5, 1e5, shape=(self.num_envs, *action_shape))

    def _convert_actions(self, actions):
        # same rule as PairScores with n = n_lines (breeding_programs_env.py:25-37)
        return PairScores._convert_actions(_LinesView(self), actions)

    def step(self, actions):
        env, sim = self.env, self.simulator
        low = self._convert_actions(actions)  # (E, n_lines, 2) indices into the current population
        if env.storage == "bits" and env._prows is not None:
            infos = self._generation_on_bit_planes(low)
        else:
            infos = self._generation_on_byte_rows(low)
        return env._close_step(infos)

    def _generation_on_byte_rows(self, low):
        """cross -> doubled haploids -> the best of every line -> the best of those, every stage materialised in the
        byte layout through the simulator's public operators (``storage="bytes"``)."""
        env, sim = self.env, self.simulator
        eng, E = sim.engine, self.num_envs
        f1 = eng.empty_population(E * self.n_lines)
        sim.cross_indexed(env._rows, E, env._n, low, self.n_lines, f1, None, env._rng_mode)
        f1 = eng.as_genotypes(f1, E, self.n_lines)
        dh = sim.double_haploid(f1, n_offspring=self.plant_per_line, rng_mode=env._rng_mode)  # (E, L, P, M, 2)
        kept, _ = sim.select(dh, k=self.k_per_line)  # (E, L, k_per_line, M, 2)
        kept = kept.reshape(E, self.n_lines * self.k_per_line, sim.n_markers, 2)
        env.populations, _ = sim.select(kept, k=self.individual_per_gen)
        return env.get_info()

    def _index_tables(self, E, L, P, K, n, dev):
        """What does not change from step to step: every plant's F1 (twice: a doubled haploid has one parent), and the
        first row of every line / env in the flat row numbering the gathers use."""
        key = (E, L, P, K, n, str(dev))
        if getattr(self, "_ix_key", None) != key:
            one = torch.arange(L, dtype=torch.int32, device=dev).repeat_interleave(P)
            self._ix = {
                "dh": one.repeat(E).reshape(E, L * P, 1).expand(-1, -1, 2).contiguous(),
                "line0": (torch.arange(E * L, dtype=torch.int32, device=dev) * P).unsqueeze(1),
                "env0": (torch.arange(E, dtype=torch.int32, device=dev) * (L * K)).unsqueeze(1),
            }
            self._ix_key = key
        return self._ix

    def _generation_on_bit_planes(self, low):
        """The same generation with every population as bit planes (``storage="bits"``, the default): the 80,000
        doubled haploids of 4 envs x 200 lines x 100 plants are 0.2 GB instead of 1.6 GB, their GEBV comes out of the
        launch that makes them (or a pass over the planes), the selections gather plane rows, and only the
        ``individual_per_gen`` survivors are expanded to the byte layout the step returns.  Same keys, same plants."""
        env, sim = self.env, self.simulator
        eng, E, L, P, K, n = sim.engine, self.num_envs, self.n_lines, self.plant_per_line, self.k_per_line, self.individual_per_gen
        dev, T = eng.device, eng.n_traits
        ix = self._index_tables(E, L, P, K, n, dev)
        hint = getattr(eng, "hint_next_cross", None)  # the launches alternate two batch shapes: say which comes next,
        dh_mode = env._rng_mode | _lib.RNG_DOUBLE_HAPLOID  # so that each makes the other's crossover masks ahead
        f1 = eng.empty_packed(E * L)
        if hint:
            hint(E, L * P, dh_mode)
        sim.cross_indexed_packed(env._prows, E, env._n, low, L, f1, None, None, env._rng_mode)
        # every F1 gives P doubled haploids (Simulator.double_haploid's key plumbing: one key per plant)
        dh, g = eng.empty_packed(E * L * P), eng.empty((E * L, P, T), torch.float32)
        if hint:
            hint(E, L, env._rng_mode)
        sim.cross_indexed_packed(f1, E, L, ix["dh"], L * P, dh, None, g, dh_mode)
        # the K best plants of every line by the conventional index (the sum of the GEBVs over traits), best first
        best = eng.topk(g.sum(dim=-1) if T > 1 else g.reshape(E * L, P), K)  # (E * L, K) int32
        flat = (best + ix["line0"]).reshape(-1)
        kept, gk = eng.empty_packed(E * L * K), eng.empty((E, L * K, T), torch.float32)
        eng.gather_packed(dh, flat, E * L * K, kept, None, gk)
        # ... and the n best of an env's kept plants are the next population
        best = eng.topk(gk.sum(dim=-1) if T > 1 else gk.reshape(E, L * K), n)  # (E, n)
        flat = (best + ix["env0"]).reshape(-1)
        newp, gn = eng.empty_packed(E * n), eng.empty((E, n, T), torch.float32)
        newb = eng.empty_population(E * n) if env.observations == "eager" else None
        eng.gather_packed(kept, flat, E * n, newp, newb, gn)
        env._adopt(newb, n, newp)
        return {"GEBV": DeviceArray.wrap(gn)}


class _LinesView:
    """What PairScores._convert_actions reads from `self`, with n_crosses = n_lines."""

    def __init__(self, w):
        self.simulator = w.simulator
        self.num_envs = w.num_envs
        self.n_crosses = w.n_lines

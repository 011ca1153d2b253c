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
import numpy as np

from .. import spaces
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
        eng = sim.engine
        E = self.num_envs
        low = self._convert_actions(actions)  # (E, n_lines, 2) indices into the current population
        f1 = eng.empty_population(E * self.n_lines)
        sim.cross_indexed(env._rows, E, env._n, low, self.n_lines, f1, None, env._rng_mode)
        f1 = eng.as_genotypes(f1, E, self.n_lines)
        dh = sim.double_haploid(f1, n_offspring=self.plant_per_line, rng_mode=env._rng_mode)  # (E, L, P, M, 2)
        kept, _ = sim.select(dh, k=self.k_per_line)  # (E, L, k_per_line, M, 2)
        kept = kept.reshape(E, self.n_lines * self.k_per_line, sim.n_markers, 2)
        env.populations, _ = sim.select(kept, k=self.individual_per_gen)

        env.step_idx += 1
        infos = env.get_info()
        done = env.step_idx == self.num_generations
        if self.reward_shaping or done:
            rews = np.max(infos["GEBV"], axis=(1, 2))
            rews = np.asarray(rews)
        else:
            rews = np.zeros(self.num_envs)

        if done and self.autoreset:
            self.reset()

        terminated = np.full(self.num_envs, False)
        truncated = np.full(self.num_envs, done)
        return self.populations, rews, terminated, truncated, infos


class _LinesView:
    """What PairScores._convert_actions reads from `self`, with n_crosses = n_lines."""

    def __init__(self, w):
        self.simulator = w.simulator
        self.num_envs = w.num_envs
        self.n_crosses = w.n_lines

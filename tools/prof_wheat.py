"""WheatBreedGym steps alone (for rocprofv3 --kernel-trace --stats): 4 envs x 200 lines x 100 plants x 10k markers."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from breedgym_amd import VecBreedGym
from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes
from breedgym_amd.vector.breeding_programs_env import WheatBreedGym

gm = synthetic_genetic_map(10_000); geno = synthetic_genotypes(370, 10_000)
w = WheatBreedGym(VecBreedGym(num_envs=4, initial_population=geno, individual_per_gen=200, genetic_map=gm, sync_rewards=False,
                              num_generations=10**6), n_lines=200, plant_per_line=100, k_per_line=5)
w.reset(seed=0)
sc = torch.randn(4, 200, 200, device="cuda")
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20): w.step(sc)
torch.cuda.synchronize()

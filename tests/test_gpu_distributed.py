"""GPU: the N > 1 paths with the HIP engine and TWO ranks -- two child processes sharing device 0, the process group
over gloo (RCCL refuses two ranks on one device; gloo moves the device tensors through the host, see
vector/distributed.py).  The workers and every assertion are tests/test_distributed.py's: N ranks == one device, row
for row, against the oracle's whole-population result -- env-sharded stepping in both key modes with the selection
wrapper's shuffles through the env window; the population-sharded generation (device-side winner layout, speculative
block size incl. the redo of a bad guess, winners by one all-gather, cross window) in both storages with cross-shard
ties.  What this cannot show is RCCL moving the bytes: that needs more than one GPU (bench.py --gpus N).

Also: bh_winner_layout against its NumPy restatement."""
import numpy as np
import pytest
import torch

from test_distributed import _env_sharded_worker, _pop_sharded_worker, _run

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(600)
@pytest.mark.parametrize("env_rng", ["shared", "independent"])
def test_env_sharded_two_ranks_on_one_gpu(env_rng):
    out = _run(_env_sharded_worker, extra=(env_rng, "hip"), join_s=420)
    assert out[0][1] == out[1][1] and out[0][2] == out[1][2]


@pytest.mark.timeout(600)
@pytest.mark.parametrize("storage", ["bytes", "bits"])
def test_population_sharded_two_ranks_on_one_gpu(storage):
    out = _run(_pop_sharded_worker, extra=(storage, "hip"), join_s=420)
    assert out[0][1] == out[1][1]


@pytest.mark.timeout(600)
def test_population_sharded_four_ranks_on_one_gpu():
    """Four ranks (four processes on device 0, gloo): winners from every shard land in every rank's gathered buffer
    through the device-side layout; the same assertions as with two."""
    out = _run(_pop_sharded_worker, world=4, extra=("bits", "hip"), join_s=420)
    assert out[0][1] == out[1][1] == out[2][1] == out[3][1]


@pytest.mark.timeout(900)
def test_population_sharded_two_ranks_at_configs3_row_length():
    """configs[3]'s rows (100,000 markers: 25,088-byte bit-plane rows on whole lines, the long-row launch with independent
    keys, the matrix-core GEBV pass, radix-select top-k) in two shards of 1,024, k = n // 10 winners per generation from
    both shards, four generations with a redone one: every rank's rows == the rows one device computes for the whole
    population (the C oracle's), byte for byte -- the toy shape's assertions where the kernels are the ones configs[3] runs."""
    out = _run(_pop_sharded_worker, extra=("bits", "hip", (2048, 100_000, 204)), join_s=800)
    assert out[0][1] == out[1][1]


@pytest.mark.timeout(900)
def test_env_sharded_two_ranks_at_the_headline_shape():
    """What ``bench.py --gpus 2`` runs per rank -- 16 envs x 256 x 10k out of a 370-row germplasm through the two-role
    launch, bit-plane state, the env window (rank x 16 of 32) -- checked where the toy shape's assertions are: founders,
    three steps and the selection wrapper's shuffles byte for byte against ONE 32-env oracle run."""
    out = _run(_env_sharded_worker, extra=("shared", "hip", (370, 10_000, 16, 256, 25, 256)), join_s=800)
    assert out[0][1] == out[1][1] and out[0][2] == out[1][2]


@pytest.mark.timeout(600)
def test_env_sharded_three_ranks_on_one_gpu():
    out = _run(_env_sharded_worker, world=3, extra=("shared", "hip"), join_s=420)
    assert out[0][1] == out[1][1] == out[2][1] and out[0][2] == out[1][2] == out[2][2]


@pytest.mark.parametrize("k,n_local,world,cap", [(1, 5, 1, 1), (9, 20, 2, 5), (9, 20, 2, 1), (819, 1024, 8, 140), (1500, 700, 3, 700),
                                                 (6553, 8192, 8, 900), (6553, 8192, 8, 50), (65536, 8192, 8, 8192),
                                                 (5000, 64, 256, 40)])
def test_winner_layout_kernel(k, n_local, world, cap, rng):
    from breedgym_amd import HipEngine
    from oracle import numpy_oracle as no

    eng = HipEngine(np.full(64, 0.01, np.float32), np.ones((64, 1), np.float32))
    winners = rng.permutation(n_local * world)[:k].astype(np.int32)
    if k > 100:  # an uneven split: most winners on two ranks
        heavy = rng.integers(0, min(2, world), k // 2) * n_local + rng.integers(0, n_local, k // 2)
        winners[: k // 2] = heavy.astype(np.int32)
    pairs = rng.integers(-k - 3, k + 3, (n_local, 2)).astype(np.int32)
    w_d, p_d = torch.from_numpy(winners).to(eng.device), torch.from_numpy(pairs).to(eng.device)
    for rank in sorted({0, world // 2, world - 1}):
        send, pos, counts, idx = eng.winner_layout(w_d, n_local, world, rank, cap, p_d)
        r_send, r_pos, r_counts, r_idx = no.winner_layout(winners, n_local, world, rank, cap, pairs)
        assert np.array_equal(counts.cpu().numpy(), r_counts), rank
        if r_counts[world] <= cap:  # everything has a place: all of it must agree
            assert np.array_equal(send.cpu().numpy(), r_send) and np.array_equal(pos.cpu().numpy(), r_pos)
            assert np.array_equal(idx.cpu().numpy(), r_idx)
        else:  # blocks too small: only the counts are promised (the caller redoes the call), but nothing may be out of range
            assert int(pos.max()) < world * cap and int(send.max()) < n_local and int(send.min()) >= 0
    send, pos, counts, idx = eng.winner_layout(w_d, n_local, world, 0, cap)
    assert idx is None and np.array_equal(counts.cpu().numpy(), no.winner_layout(winners, n_local, world, 0, cap)[2])


def _bench(*args, env_extra=None, timeout=800):
    """``python bench.py ...`` as the driver spells it: plain python, no launcher, no RANK in the environment."""
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, str(root / "bench.py"), *args], env=env, capture_output=True, text=True, timeout=timeout)


def _one_line(r):
    import json

    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


@pytest.mark.timeout(900)
@pytest.mark.parametrize("mode", ["envs", "population"])
def test_bench_self_launch_two_ranks_on_one_gpu(mode):
    """``python bench.py --gpus 2 [--mode population]`` as the driver spells it (plain python, no launcher): the parent
    starts two fresh child ranks before touching the GPU and relays rank 0's line.  One GPU here, so ``--oversubscribe``
    (gloo): the line must say what the ranks saw and that it is a REHEARSAL of the code path, not a measurement."""
    steps = ("--steps", "20", "--warmup", "5", "--long-steps", "200") if mode == "envs" else ("--steps", "3", "--warmup", "1")
    line = _one_line(_bench("--gpus", "2", "--oversubscribe", *steps, "--headline-only", "--mode", mode))
    d = line["distributed"]
    assert line["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["world_size"] == 2 and "gloo" in d["backend"]
    assert d["launcher"] == "bench.py self-launch" and "oversubscribed" in d
    assert 0 < d["rank_seconds_min"] <= d["rank_seconds_max"] and line["value"] > 0
    if mode == "envs":
        assert line["value_long"] > 0 and line["config"]["crosses_per_step_per_gpu"] == 4096
    else:  # BASELINE configs[3] as `--gpus N --mode population` runs it: 8192 x 100k per rank
        assert line["generations_redone"] >= 0 and line["config"]["crosses_per_step_per_gpu"] == 8192
        assert "16384 individuals" in line["config"]["workload"] and line["scaling"] == "weak"


@pytest.mark.timeout(900)
def test_bench_self_launch_four_ranks_on_one_gpu():
    """A wider rehearsal than two ranks, inside what one GPU box allows: its process guard admits six processes on the
    card, this test process is one of them and whatever runs the suite may be another -- so four ranks, not the eight the
    driver runs on an 8-GPU node (five have run too: profiles/r05_soak.txt).  Env-sharded, the env window over 4 x 16
    envs; every rank must be seen."""
    line = _one_line(_bench("--gpus", "4", "--oversubscribe", "--headline-only", "--steps", "5", "--warmup", "2", "--long-steps", "0"))
    d = line["distributed"]
    assert line["n_gpus"] == 4 and d["ranks_seen"] == 4 == d["world_size"] and "oversubscribed" in d
    assert line["value"] > 0 and "value_long" not in line


@pytest.mark.timeout(600)
def test_bench_self_launch_exits_nonzero_when_a_rank_is_killed():
    """A rank that dies once the process group is up (SIGKILL: no exception, no goodbye) must end the job: the launcher
    watches every child, ends the survivors -- who would otherwise wait in a collective for the backend's timeout -- and
    exits non-zero without a JSON line."""
    import time

    t0 = time.monotonic()
    r = _bench("--gpus", "2", "--oversubscribe", "--headline-only", "--steps", "5", "--warmup", "2", "--long-steps", "0",
               env_extra={"BENCH_TEST_KILL_RANK": "1"}, timeout=500)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")], r.stdout
    assert "rank exit codes" in r.stderr and "-9" in r.stderr, r.stderr[-1500:]
    assert time.monotonic() - t0 < 400


@pytest.mark.timeout(900)
@pytest.mark.parametrize("mode", ["envs", "population"])
def test_bench_under_torchrun_with_rccl_one_rank(mode):
    """The driver's spelling for N > 1 -- ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N ...`` -- with the one rank a one-GPU box allows: the process group comes
    up over RCCL (backend "nccl", device_id given), the timing all-reduces and -- in population mode -- the all-gathers
    of candidates and of the winners' rows go through RCCL.  One rank moves no bytes between GPUs; what this shows is
    that the RCCL code path of both modes executes (rounds 1-4 never ran it)."""
    import json
    import os
    import socket
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "BENCH_BACKEND")}
    steps = ("--steps", "20", "--warmup", "5", "--long-steps", "100", "--headline-only") if mode == "envs" else ("--steps", "3", "--warmup", "1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), str(root / "bench.py"), "--gpus", "1", *steps, "--mode", mode],
                       env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    d = line["distributed"]
    assert d["backend"] == "nccl (RCCL)" and d["ranks_seen"] == 1 == d["world_size"] and d["launcher"] == "torch.distributed.run"
    assert "oversubscribed" not in d and line["n_gpus"] == 1 and line["value"] > 0
    if mode == "population":
        assert line["generations_redone"] == 0 and "8192 individuals" in line["config"]["workload"]

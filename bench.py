#!/usr/bin/env python
"""bench.py -- crosses/s through vector env.step() on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode envs|population]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Both spellings work for N > 1: started as plain ``python bench.py --gpus N`` (no RANK in the environment) this
process becomes the launcher -- it starts N fresh child ranks of itself BEFORE anything here touches the GPU (child
processes, never an exec), relays rank 0's JSON line and exits non-zero if any rank does.  Ranks talk over RCCL
(backend "nccl"), one rank per GPU.  ``--oversubscribe`` lets N exceed the GPUs of the box: ranks then share devices
and the process group runs over gloo (RCCL refuses two ranks per device) -- a rehearsal of the N-rank code path on one
GPU, marked as such in the line, never a scaling number.

Workload (N = 1, default mode): BASELINE.json configs[1] -- a 370-individual x 10k-marker synthetic
germplasm, 16 envs x 256 individuals = 4096 crosses per generation, GEBV reward every step, episodes
of 10 generations with autoreset (so every 10th step also pays a reset, as in the reference's
scripts/time_vec.py:57-65 protocol).  A "step" is one ``VecBreedGym.step``: fused parent gather +
meiosis + offspring write + GEBV + reward, and the reset when due.  Inputs (germplasm, genetic map,
actions) are resident in HBM before the timed region; outputs stay on the device
(``sync_rewards=False``), nothing is copied to the host inside it.

N > 1, ``--mode envs`` (default): one process per GPU, every rank steps its own 16 envs (env-sharded,
SURVEY.md 8e): no data-path collective, "weak" scaling; value = all ranks' crosses / max rank time.
``--mode population``: BASELINE configs[3] -- one population of N x 8192 individuals x 100k markers
sharded row-wise over the ranks; a step is one generation of ``ShardedPopulation.select_and_cross``
(local GEBV -> local top-k -> RCCL all-gather of candidates -> global top-k -> RCCL all-gather of the
winners' bit-plane rows -> every rank crosses its shard); weak scaling as well (8192 per rank).

The JSON line also carries, at N = 1: ``roofline`` (dominant kernel of the headline: bytes it must
move for the storage in use / its HIP-event time / 8 TB/s), ``configs`` (configs[2] and configs[4]
timed the same way), ``cpu_baseline`` (the C oracle on the host's cores).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

N_GERMPLASM, N_MARKERS, N_ENVS, N_PER_ENV, N_GENERATIONS = 370, 10_000, 16, 256, 10
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def packed_pitch(M):
    return 16 * ((M + 63) // 64)


def cross_bytes(M, C, B, storage, observations):
    """Bytes one cross launch must move for the layout in use:
    bits  : per cross two bit-plane parents read (2 x M/4), the bit-plane offspring written (M/4), the
            byte-layout observation written when eager (2M); per keyed cross (B of them: n with shared
            keys) one mask row read and the next generation's written (M/4 each);
    bytes : per cross two byte-layout parents read (2 x 2M) and the offspring written (2M), + masks.
    Shared, cache-resident tables (thresholds, effect digits) are not counted."""
    pp = packed_pitch(M)
    masks = 2 * B * pp
    if storage == "bits":
        return C * (3 * pp + (2 * M if observations == "eager" else 0)) + masks
    return C * 6 * M + masks


def cpu_baseline(gm, geno, rng_mode, seconds_hint=12.0):
    """The CPU oracle (a from-scratch port: the reference's chromax/JAX CPU path cannot be
    installed here or on the GPU box) timed on this host on a bounded sample of the same
    workload: whole 4096-cross generations of 370x10k (cross + GEBV + reward), OpenMP over
    crosses on the box's CPU share (16 hardware threads per GPU)."""
    from oracle import c_oracle as co
    from oracle import numpy_oracle as no

    rng = np.random.default_rng(0)
    E, n, M = N_ENVS, N_PER_ENV, N_MARKERS
    pool = geno.view(np.uint8)[np.stack([rng.permutation(N_GERMPLASM)[:n] for _ in range(E)])]
    idx = rng.integers(0, n, (E, n, 2), dtype=np.int32)
    key = co.key_from_seed(0)
    cores = min(16, len(os.sched_getaffinity(0)))
    co.set_num_threads(cores)

    def one_generation():
        out, _ = co.cross(pool, idx, gm.recombination_vec, key, rng_mode)
        g = co.gebv(out, gm.marker_effects)
        co.reward_max(g)

    t0 = time.perf_counter()
    one_generation()  # also the warm-up
    first = time.perf_counter() - t0
    reps = int(max(1, min(2000, seconds_hint / max(first, 1e-3))))  # ~12 s of 16-thread CPU work
    t0 = time.perf_counter()
    for _ in range(reps):
        one_generation()
    dt = (time.perf_counter() - t0) / reps
    res = {
        "value": E * n / dt, "unit": "crosses/s", "cores": cores, "kind": "port",
        "sample": f"{reps} generations of {E}x{n}={E*n} crosses x {M} markers (C oracle, OpenMP, "
                  f"cross+GEBV+reward, same RNG sharing as the GPU run)",
    }
    # single-thread C and the unfused NumPy restatement on one env's 256 crosses
    co.set_num_threads(1)
    t0 = time.perf_counter()
    out1, _ = co.cross(pool[:1], idx[:1], gm.recombination_vec, key, 1)
    co.gebv(out1, gm.marker_effects)
    res["c_oracle_1thread_crosses_per_s"] = n / (time.perf_counter() - t0)
    co.set_num_threads(cores)
    par = pool[0][idx[0]]  # materialised (n, 2, M, 2), as the reference does
    t0 = time.perf_counter()
    o, _ = no.cross_batched(par, gm.recombination_vec, key)
    no.gebv(o, gm.marker_effects)
    res["numpy_restatement_crosses_per_s"] = n / (time.perf_counter() - t0)
    return res


SETTLE_S = 0.1  # untimed steps before the warm-up: the GPU's clocks take tens of ms under load to settle (a 200-step
                # sample right after start-up reads 27 us per step, the same loop 15 ms later 25)


def time_steps(env, actions, steps, warmup, barrier):
    """-> (seconds for `steps` steps incl. the final sync, seconds the host needed to issue them)."""
    import torch

    t_end = time.perf_counter() + SETTLE_S
    i = 0
    while time.perf_counter() < t_end:
        env.step(actions[i % len(actions)])
        i += 1
        if i % 64 == 0:
            torch.cuda.synchronize()  # keep the queue short: the wall clock, not the backlog, ends the settling
    for i in range(warmup):
        env.step(actions[i % len(actions)])
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        env.step(actions[i % len(actions)])
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    barrier()
    return dt, t_issue


def time_events(launch, reps, warm=5):
    """HIP events on the stream the launches go to (torch's current stream = the engine's), around
    `reps` back-to-back calls of `launch(i)`; -> ms per call."""
    import torch

    for i in range(warm):
        launch(i)
    # ... and the clocks settled as in time_steps: the same untimed SETTLE_S of launches before the events (a 200-launch
    # sample taken right behind a synchronisation read 19.4 ... 22.7 us for the same launch from run to run)
    t_end, i = time.perf_counter() + SETTLE_S, warm
    while time.perf_counter() < t_end:
        launch(i)
        i += 1
        if i % 64 == 0:
            torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    ev0.record()
    for i in range(reps):
        launch(i)
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / reps


def time_cross_launches(env, actions, reps, with_gebv=True, own_path=True):
    """The hot-path launch alone (the fused two-role launch: streams this generation, makes the next
    one's masks), in the env's own storage layout, through the same C entry point the env uses but
    without the reward reduction and the reset.  ``with_gebv=False`` leaves the GEBV out: where it is a
    pass of its own (long rows, three traits and more) that isolates the cross launch."""
    import torch

    eng = env.simulator.engine
    E, n = env.num_envs, env._n
    gebv = eng.empty((E, n, eng.n_traits), torch.float32) if with_gebv else None
    key = np.ascontiguousarray(env.simulator.random_key.copy())
    ring = getattr(env, "_ring", None)
    if (own_path and env.storage == "bits" and with_gebv and ring is not None and len(ring.state) >= 2
            and (ring.n_envs, ring.n_out) == (E, n) and env._prows is not None):
        # the env's own launch path on the env's own blocks: StepRing.step (one bh_step call) into whichever slot is
        # free, the slot just written being the next call's parents -- what a loop that keeps nothing does, without the
        # env's bookkeeping, the reward and the resets.  (Through eng.cross_packed on freshly allocated blocks the same
        # launch read 19.4 ... 22.7 us: that wrapper's host side is as long as the launch, and where the blocks land
        # matters.)  The env is not stepped again after this.
        cur = [env._prows]
        flags = env._rng_mode | (ring.OBS if env.observations == "eager" else 0)

        def launch(i):
            slot, _ = ring.step(cur[0].data_ptr(), n, actions[i % len(actions)], key, flags, False, False)
            cur[0] = ring.state[slot]
    elif env.storage == "bits":
        pool, outp = env._prows, eng.empty_packed(E * n)
        obs = eng.empty_population(E * n) if env.observations == "eager" else None

        def launch(i):
            eng.cross_packed(pool, E, n, actions[i % len(actions)], n, key, env._rng_mode, outp, obs, gebv)
    else:
        pool, out = env._rows, eng.empty_population(E * n)

        def launch(i):
            eng.cross(pool, E, n, actions[i % len(actions)], n, key, env._rng_mode, out, gebv)

    return time_events(launch, reps)


def roofline_of(kernel, nbytes, ms, **extra):
    achieved = nbytes / (ms * 1e-3) / 1e9
    if extra.get("traffic") is not None:
        # what the counters are (MI355X_MICROARCH.md, HBM section): requests at the memory side of the L2, Infinity-Cache
        # hits included -- an upper bound on HBM bytes, not HBM bytes; the 2 x on FETCH_SIZE is calibrated for the
        # 16-byte-per-lane streaming reads these kernels issue (LDS-DMA and global_load alike)
        extra.setdefault("traffic_is", "memory-side (fabric) bytes per launch: 2 x FETCH_SIZE + WRITE_SIZE; Infinity-Cache hits "
                                       "are counted, so at cache-resident sizes this is an upper bound on HBM bytes")
    return {"kernel": kernel, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS, "bytes_per_launch": nbytes, "kernel_ms": ms, **extra}


_LIVE_TRAFFIC = {}  # kernel name (as rocprofv3 prints it, blanks removed) -> HBM bytes per launch, measured by this run
_LIVE_NOTE = None


def traffic_child():
    """Child of live_traffic(), run under ``rocprofv3 --pmc``: exactly the launches the roofline objects time
    (time_cross_launches of the headline env and of configs[2], the two GEBV routes of configs[4]), nothing else."""
    import torch

    from breedgym_amd import HipEngine, VecBreedGym
    from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes

    torch.cuda.set_device(0)
    for (N, M, E, n, chrs, reps, with_gebv) in ((N_GERMPLASM, N_MARKERS, N_ENVS, N_PER_ENV, 10, 30, True),
                                                (2048, 100_000, 4, 2048, 21, 6, False)):
        gm = synthetic_genetic_map(M, n_chr=chrs, n_traits=1, seed=1)
        env = VecBreedGym(num_envs=E, initial_population=synthetic_genotypes(N, M, seed=0), individual_per_gen=n,
                          num_generations=N_GENERATIONS, autoreset=True, reward_shaping=True, sync_rewards=False, genetic_map=gm)
        arng = np.random.default_rng(2)
        actions = [torch.from_numpy(arng.integers(0, n, (E, n, 2), dtype=np.int32)).to(env.device) for _ in range(4)]
        env.reset(seed=100)
        time_cross_launches(env, actions, reps, with_gebv=with_gebv)
        del env
    N, M, T = 2048, 100_000, 8
    gm = synthetic_genetic_map(M, n_chr=21, n_traits=T, seed=1)
    eng = HipEngine(gm.recombination_vec, gm.marker_effects)
    rows = eng.upload_population(synthetic_genotypes(N, M, seed=0))
    packed = eng.pack(rows)
    g = eng.empty((N, T), torch.float32)
    for _ in range(8):
        eng.gebv(rows, g)
        eng.gather_packed(packed, None, N, None, None, g)
    torch.cuda.synchronize()


def live_traffic(timeout=90):
    """HBM-side bytes per launch of the hot kernels, measured by THIS run: two child processes of this script under
    ``rocprofv3 --kernel-trace --pmc FETCH_SIZE`` and ``... --pmc WRITE_SIZE`` (separate passes, the program itself
    after ``--``; /opt/skills/guides/MI355X_MICROARCH.md, HBM section), started before this process touches the GPU.
    Correction as the guide prescribes for gfx950: both counters are KiB; FETCH_SIZE reports half the bytes of wide
    streaming reads -> 2 x FETCH_SIZE + WRITE_SIZE.  Any failure leaves the committed passes as the (named) source."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    from collections import defaultdict

    global _LIVE_NOTE
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        _LIVE_NOTE = "rocprofv3 not found"
        return
    tmp = tempfile.mkdtemp(prefix="bh_pmc_", dir="/tmp")
    per = {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = [rocprof, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", f"{tmp}/{ctr}", "-o", "p", "--",
                   sys.executable, str(ROOT / "bench.py"), "--traffic-child"]
            # own process group: on a timeout the profiler AND the python child under it are ended, so that nothing of
            # them overlaps the timed region
            proc = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL,
                                    stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                proc.wait(timeout=timeout)
            except subprocess.TimeoutExpired:
                os.killpg(proc.pid, 9)
                proc.wait()
                _LIVE_NOTE = f"rocprofv3 --pmc {ctr} pass timed out after {timeout} s"
                return
            r = proc
            files = glob.glob(f"{tmp}/{ctr}/**/*counter_collection.csv", recursive=True)
            if r.returncode != 0 or not files:
                _LIVE_NOTE = f"rocprofv3 --pmc {ctr} pass failed (rc {r.returncode})"
                return
            acc = defaultdict(list)
            for row in csv.DictReader(open(files[0])):
                if row["Counter_Name"] == ctr:
                    acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
            per[ctr] = {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}
        for k, (f, nf) in per["FETCH_SIZE"].items():
            w = per["WRITE_SIZE"].get(k, (0.0, 0))[0]
            name = k.split("(")[0].replace("void ", "").replace("bh::", "").replace(" ", "")
            _LIVE_TRAFFIC[name] = ((fetch_factor(name) * f + w) * 1024.0, nf)
        _LIVE_NOTE = "ok"
    except Exception as e:  # a missing counter, a timeout: the committed passes remain
        _LIVE_NOTE = f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def fetch_factor(kernel_name):
    """What FETCH_SIZE has to be multiplied by.  The guide's gfx950 correction (x 2) is for wide coalesced streaming
    reads, which the counter tallies at half their bytes.  gebv_bits_kernel's row reads are not of that shape -- eight
    lanes take the eight 16-byte pieces of a 128-byte line in a permuted order (LDS-DMA, csrc/gebv_bits.h) -- and are
    tallied in full: with the digit-operand DMAs removed the counter reads 59.6 MB per launch for 51.2 MB of rows,
    with the row DMAs removed 3.9 MB for 6.4 MB of operands (profiles/r04_ab_gebv.txt, section 5).  So for that kernel
    the counter is taken as it is (the operands' other half, 4 MB at 2048 x 100k x 8, is left out: -5 %)."""
    return 1.0 if "gebv_bits_kernel" in kernel_name else 2.0


def committed_traffic(kernel_prefix):
    """PMC traffic (2 x FETCH_SIZE + WRITE_SIZE, per launch): measured by this run (live_traffic) where that worked,
    else from the committed rocprofv3 passes under profiles/ -- a measurement of an earlier run of the same command,
    NOT of this run.  The source is named either way."""
    want = kernel_prefix.replace(" ", "")
    for name, (b, n) in _LIVE_TRAFFIC.items():
        if name.startswith(want):
            return b, (f"measured by this run: child passes under rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE "
                       f"(separate), mean of {n} launches of {name}, 2 x FETCH_SIZE + WRITE_SIZE in KiB")
    for name in ("r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "pmc_traffic.json"):
        f = ROOT / "profiles" / name
        if not f.exists():
            continue
        try:
            d = json.loads(f.read_text())
        except Exception:
            continue
        entries = d.get("kernels", [d])
        for e in entries:
            if str(e.get("kernel", "")).replace(" ", "").startswith(kernel_prefix.replace(" ", "")):
                return e.get("hbm_bytes_per_launch"), (f"profiles/{name} (rocprofv3 --pmc passes of an earlier run; not measured "
                                                       f"by this run: {_LIVE_NOTE or 'live passes not requested'})")
    return None, "none: no committed PMC pass for this kernel"


def bench_c3(steps, warmup, barrier, variants=("eager", "lazy_observations")):
    """BASELINE configs[2]: 2048 x 100k germplasm, 4 envs x 2048 = 8192 crosses per step, defaults."""
    import torch

    from breedgym_amd import VecBreedGym
    from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes

    N, M, E, n = 2048, 100_000, 4, 2048
    gm = synthetic_genetic_map(M, n_chr=21, n_traits=1, seed=1)
    geno = synthetic_genotypes(N, M, seed=0)
    out = {"workload": "BASELINE configs[2]: 2048 x 100k germplasm, 4 envs x 2048 = 8192 crosses/gen, GEBV reward "
                       "every step, 10-generation episodes with autoreset, shared keys, bit-plane state"}
    arng = np.random.default_rng(3)
    for name, obs in (("eager", "eager"), ("lazy_observations", "lazy")):
        if name not in variants:
            continue
        env = VecBreedGym(num_envs=E, initial_population=geno, individual_per_gen=n, num_generations=N_GENERATIONS,
                          autoreset=True, reward_shaping=True, sync_rewards=False, genetic_map=gm, observations=obs)
        actions = [torch.from_numpy(arng.integers(0, n, (E, n, 2), dtype=np.int32)).to(env.device) for _ in range(4)]
        env.reset(seed=7)
        dt, _ = time_steps(env, actions, steps, warmup, barrier)
        # (both through eng.cross_packed on blocks of their own: the GEBV pass is the difference of the two)
        kms_all = time_cross_launches(env, actions, max(10, steps), own_path=False)
        kms = time_cross_launches(env, actions, max(10, steps), with_gebv=False, own_path=False)
        nbytes = cross_bytes(M, E * n, n, "bits", obs)
        r = {"value": steps * E * n / dt, "unit": "crosses/s", "ms_per_step": dt / steps * 1e3,
             "roofline": roofline_of("packed_long_kernel<1,8> + mask_finish4_kernel (long rows: four row waves stream rows "
                                     "without GEBV, eight mask waves draw the next generation's sites one gamete stream "
                                     "per lane -- 2048 keyed crosses x 100k markers of Threefry; then site words -> masks)",
                                     nbytes + 2 * 4 * n * (M // 128 + 2) * 8, kms,  # + site words written and read back
                                     api_level_GBps=6 * M * E * n / (kms * 1e-3) / 1e9),
             # both kernels of the timed pair; the PMC passes are of the eager configuration
             "traffic": (committed_traffic("packed_long_kernel")[0] or 0) + (committed_traffic("mask_finish")[0] or 0)
                        if obs == "eager" and committed_traffic("packed_long_kernel")[0] else None,
             "traffic_source": committed_traffic("packed_long_kernel")[1] if obs == "eager" else None,
             "gebv_pass_ms": kms_all - kms,
             "gebv_pass_note": "gebv_bits_kernel<1,4> over the bit-plane offspring (0.2 GB); whoever completes a cell writes its GEBV",
             "step_frac_of_hbm_peak": (nbytes + E * n * packed_pitch(M)) / (dt / steps) / 1e9 / HBM_PEAK_GBPS}
        if name == "eager":
            out.update(r)
        else:
            out[name] = r
        del env
    return out


def bench_c5(reps):
    """BASELINE configs[4]: GEBV of 2048 individuals x 100k markers x 8 traits (int8 digits on the matrix cores)."""
    import torch

    from breedgym_amd import HipEngine
    from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes

    N, M, T = 2048, 100_000, 8
    gm = synthetic_genetic_map(M, n_chr=21, n_traits=T, seed=1)
    geno = synthetic_genotypes(N, M, seed=0)
    eng = HipEngine(gm.recombination_vec, gm.marker_effects)
    rows = eng.upload_population(geno)
    packed = eng.pack(rows)
    g = eng.empty((N, T), torch.float32)
    ms_bytes = time_events(lambda i: eng.gebv(rows, g), reps)
    ms_bits = time_events(lambda i: eng.gather_packed(packed, None, N, None, None, g), reps)
    eff_bytes = M * 64  # the int8 digit pack, read once per row tile from L2
    return {
        "workload": "BASELINE configs[4]: GEBV of 2048 x 100k markers x 8 traits, exact int8-digit products on MFMA",
        "byte_rows": {"ms": ms_bytes, "individuals_per_s": N / (ms_bytes * 1e-3),
                      "roofline": roofline_of("gebv_bytes_kernel<2> (bh_gebv, byte-layout rows in one pass: bit planes built in LDS, digit GEMMs on the matrix cores, 64-bit sums by atomics)", N * 2 * M + eff_bytes, ms_bytes,
                                              traffic=committed_traffic("gebv_bytes_kernel<2")[0], traffic_source=committed_traffic("gebv_bytes_kernel<2")[1])},
        "bit_planes": {"ms": ms_bits, "individuals_per_s": N / (ms_bits * 1e-3),
                       "roofline": roofline_of("gebv_bits_kernel<2,4> (bh_gather_packed, bit-plane rows: rows and digit operands by LDS-DMA, two row tiles per operand read)",
                                               N * packed_pitch(M) + eff_bytes, ms_bits,
                                               traffic=committed_traffic("gebv_bits_kernel<2")[0],
                                               traffic_source=committed_traffic("gebv_bits_kernel<2")[1],
                                               note="51 MB of rows: not bound by its bytes.  Ramp, first operands and the way out of the K slices "
                                                    "are ~14 of its ~29 us; the stream between runs at ~55 % of the matrix cores' rate with VALU "
                                                    "(bits -> int8 operands), LDS and MFMA each busy more than half of it (profiles/r04_ab_gebv.txt)",
                                               mfma_int8_Tops=2.0 * N * M * 64 / (ms_bits * 1e-3) / 1e12, mfma_int8_peak_Tops=5000.0),
                       "api_level_GBps": N * 2 * M / (ms_bits * 1e-3) / 1e9},
    }


def population_generation(steps, warmup, rank, world, local_rank, use_dist, barrier, all_reduce=None, seen=None):
    """BASELINE configs[3]: one population sharded row-wise, 8192 x 100k per rank, k = n_total // 10.  Returns the
    JSON line of the population mode (rank 0; None elsewhere)."""
    import torch
    import torch.distributed as dist

    from breedgym_amd.simulator import Simulator
    from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes
    from breedgym_amd.vector.distributed import ShardedPopulation

    n_local, M = 8192, 100_000
    n_total = n_local * world
    k = n_total // 10
    gm = synthetic_genetic_map(M, n_chr=21, n_traits=1, seed=1)
    sim = Simulator(genetic_map=gm, seed=5, device=local_rank)
    eng = sim.engine
    base = synthetic_genotypes(512, M, seed=10 + rank)  # a rank's shard: 8192 rows drawn from 512 distinct ones
    rows = eng.upload_population(base)
    pick = torch.from_numpy(np.random.default_rng(rank).integers(0, 512, n_local, dtype=np.int32)).to(eng.device)
    shard = eng.empty_population(n_local)
    eng.gather(rows, pick, shard)
    del rows
    sp = ShardedPopulation(sim, shard, storage="bits")
    del shard
    # every rank's rows of one global pair table, the same key on every rank: the N-rank generation is the
    # generation one device would compute for the whole population (world-size invariant)
    prng = np.random.default_rng(3)
    mine = slice(rank * n_local, (rank + 1) * n_local)
    pairs = [torch.from_numpy(prng.integers(0, k, (n_total, 2), dtype=np.int32)[mine].copy()).to(eng.device) for _ in range(2)]
    key = np.array([0, 5], np.uint32)
    for i in range(warmup):
        sp.select_and_cross(k, pairs[i % 2], key)
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        sp.select_and_cross(k, pairs[i % 2], key)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    barrier()
    t = torch.tensor([dt], dtype=torch.float64, device=eng.device)
    if use_dist:
        all_reduce(t, dist.ReduceOp.MAX)
    dt_max = float(t.item())
    distributed = seen(dt) if seen is not None else None
    if rank != 0:
        return None
    pp = packed_pitch(M)
    return {
        **({"distributed": distributed} if distributed else {}),
        "generations_redone": sp.redone,  # winners' blocks guessed too small (vector/distributed.py): 0 in a steady run
        "metric": "crosses/sec (population-sharded generation: top-k GEBV selection + crosses) at 8192 x 100k markers per GPU",
        "value": world * steps * n_local / dt_max, "unit": "crosses/s", "n_gpus": world, "steps": steps,
        "warmup": warmup, "ms_per_step": dt_max / steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {
            "workload": f"BASELINE configs[3]: {n_total} individuals x 100k markers over {world} GPU(s), 8192 per rank, "
                        f"top-{k} by GEBV, every rank crosses its shard from the replicated winners",
            "n_markers": M, "crosses_per_step_per_gpu": n_local, "storage": "bits",
            "collectives": f"all-gather of {min(k, n_local)} (score, index) candidates per rank; one all-gather of the "
                           f"winners' bit-plane rows, every rank sending the rows it owns padded to the largest share "
                           f"({k * pp / 1e6:.0f} MB of rows in all)",
        },
        "bytes_per_generation_per_gpu": {"shard_read_for_gebv": n_local * pp, "winner_rows_gathered": k * pp,
                                         "parents_read": 2 * n_local * pp, "offspring_written": n_local * pp,
                                         "masks_written_and_read": 2 * n_local * pp},
        "bound": "VALU: every cross owns its keys (one flat chromax batch), 2 gametes x 50k Threefry-2x32-20 calls each -- "
                 "JAX's bit stream itself; the bit-plane traffic above is ~0.05 of what HBM moves in that time",
    }


def bench_population(args, rank, world, local_rank, use_dist, barrier, all_reduce, seen):
    line = population_generation(args.steps, args.warmup, rank, world, local_rank, use_dist, barrier, all_reduce, seen)
    if line is not None:
        print(json.dumps(line), flush=True)


def bench_c4_one_rank(steps, warmup, barrier):
    """BASELINE configs[3] as far as one GPU goes: one rank's generation (8192 x 100k shard, its own top-k), and the
    action shaping of the whole 65,536-individual population (`SelectionScores` with the reference's default
    k = n // 10: 21.5 M diallel pairs shuffled by three rounds of radix sorts in HBM)."""
    import torch

    from breedgym_amd import HipEngine

    line = population_generation(steps, warmup, 0, 1, torch.cuda.current_device(), False, barrier)
    out = {"workload": line["config"]["workload"], "ms_per_generation": line["ms_per_step"], "value": line["value"],
           "unit": line["unit"], "bytes_per_generation": line["bytes_per_generation_per_gpu"], "bound": line["bound"],
           "note": "ONE rank's share, measured on one GPU; the 8-rank run (bench.py --gpus 8 --mode population) adds the two "
                   "all-gathers over RCCL, which no 1-GPU run can time"}
    eng = HipEngine(np.full(64, 0.01, np.float32), np.ones((64, 1), np.float32))
    n, k, nc = 65_536, 6_553, 8_192
    scores = torch.randn((1, n), device=eng.device)
    key = np.array([0, 9], np.uint32)
    ms = time_events(lambda i: eng.selection_actions(scores, k, nc, key), max(20, steps), warm=3)
    out["selection_actions"] = {"shape": f"n = {n}, k = {k} (n // 10), {k * (k - 1) // 2} pairs, n_crosses = {nc}", "ms": ms,
                                "note": "bh_selection_actions: top-k through the radix sort, then the wanted-only shuffle (the 8,192 wanted "
                                        "positions propagated backwards through JAX's 3 rounds: per round two Threefry sweeps over the "
                                        "21.5 M positions, nothing sorted; csrc/shuffle_sparse.h), un-ranking"}
    return out


def self_launch(args):
    """``python bench.py --gpus N`` without a launcher: start the N ranks as fresh child processes (this process has
    not touched the GPU and never will), wait, relay rank 0's one JSON line.  -> exit code."""
    import socket
    import subprocess

    import torch  # counting devices does not initialise the GPU (no context is created)

    n = args.gpus
    ndev = torch.cuda.device_count()
    if ndev == 0:
        print("bench.py needs a GPU: breedgym_amd has no CPU fallback", file=sys.stderr)
        return 2
    backend = "nccl"
    if n > ndev:
        if not args.oversubscribe:
            print(f"--gpus {n} but this box has {ndev} GPU(s); --oversubscribe rehearses the {n}-rank path with ranks "
                  f"sharing devices over gloo (not a scaling measurement)", file=sys.stderr)
            return 2
        backend = "gloo"
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    argv = [sys.executable, str(ROOT / "bench.py")] + [a for a in sys.argv[1:]]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r % ndev), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BENCH_BACKEND=backend, BENCH_SELF_LAUNCHED="1",
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        # rank 0's stdout carries the line; the other ranks' goes to stderr
        procs.append(subprocess.Popen(argv, env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr.fileno()))
    out, _ = procs[0].communicate()
    codes = [procs[0].returncode]
    for p in procs[1:]:
        try:
            codes.append(p.wait(timeout=120))
        except subprocess.TimeoutExpired:  # a rank that outlives rank 0 by minutes is stuck: end it (its own pid)
            p.kill()
            codes.append(p.wait())
    lines = [ln for ln in out.decode().splitlines() if ln.startswith("{")]
    if any(codes) or not lines:
        sys.stderr.write(out.decode())
        print(f"bench.py: rank exit codes {codes}", file=sys.stderr)
        return 1
    print(lines[-1], flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--mode", choices=("envs", "population"), default="envs")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU leg (profiling runs)")
    ap.add_argument("--headline-only", action="store_true",
                    help="only the headline configuration: no CPU leg, no other key mode / storage / config lines "
                         "(profiling runs, so that a kernel name in the profile is one configuration)")
    ap.add_argument("--only", choices=("c3", "c3lazy", "c5", "c4"), default=None,
                    help="run only that extra configuration (profiling runs: one configuration per kernel name)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not spawn the rocprofv3 --pmc child passes behind roofline.traffic (committed passes are named instead)")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="allow more ranks than GPUs (ranks share devices, gloo instead of RCCL): a rehearsal of the "
                         "N-rank path, not a measurement")
    ap.add_argument("--long-steps", type=int, default=2000,
                    help="further timed steps behind the --steps sample, reported as value_long (0: none)")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.traffic_child:
        traffic_child()
        return

    if "RANK" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))  # before this process imports torch.cuda or touches the GPU
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    # HBM traffic of the hot kernels from PMC counters, by child processes, BEFORE this process initialises the GPU
    if (world == 1 and args.mode == "envs" and not args.headline_only and args.only is None and not args.no_live_traffic
            and "RANK" not in os.environ):
        live_traffic()

    import torch
    import torch.distributed as dist

    from breedgym_amd import VecBreedGym
    from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: breedgym_amd has no CPU fallback")
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    use_dist = "RANK" in os.environ  # launched by torch.distributed.run or by self_launch (also with one rank)
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL prints a version banner on stdout when its first communicator comes up: keep stdout for the one JSON line
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group("gloo")
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def all_reduce(t, op):  # a few numbers on the device; gloo takes them through the host
        if use_dist and backend == "nccl":
            dist.all_reduce(t, op=op)
        elif use_dist:
            h = t.cpu()
            dist.all_reduce(h, op=op)
            t.copy_(h)
        return t

    def what_the_ranks_saw(dt):
        """-> the 'distributed' object of the line: who took part and how long each rank's timed region was."""
        if not use_dist:
            return None
        ones = all_reduce(torch.ones(1, dtype=torch.float64, device=f"cuda:{local_rank}"), dist.ReduceOp.SUM)
        lo = all_reduce(torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}"), dist.ReduceOp.MIN)
        hi = all_reduce(torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}"), dist.ReduceOp.MAX)
        d = {"backend": "nccl (RCCL)" if backend == "nccl" else "gloo (device tensors staged through the host)",
             "ranks_seen": int(ones.item()), "world_size": world, "devices_on_box": torch.cuda.device_count(),
             "launcher": "bench.py self-launch" if os.environ.get("BENCH_SELF_LAUNCHED") else "torch.distributed.run",
             "rank_seconds_min": float(lo.item()), "rank_seconds_max": float(hi.item())}
        if world > torch.cuda.device_count():
            d["oversubscribed"] = (f"{world} ranks share {torch.cuda.device_count()} GPU(s): a rehearsal of the code path; "
                                   f"the value below is NOT a {world}-GPU measurement")
        return d

    if args.mode == "population":
        bench_population(args, rank, world, local_rank, use_dist, barrier, all_reduce, what_the_ranks_saw)
        if use_dist:
            dist.destroy_process_group()
        return
    if args.only in ("c3", "c3lazy"):
        v = ("eager",) if args.only == "c3" else ("lazy_observations",)
        print(json.dumps({"configs": {"c3": bench_c3(max(10, args.steps // 10), max(3, args.warmup // 10), barrier, v)}}))
        return
    if args.only == "c5":
        print(json.dumps({"configs": {"c5": bench_c5(max(20, args.steps // 4))}}))
        return
    if args.only == "c4":
        print(json.dumps({"configs": {"c4_one_rank": bench_c4_one_rank(max(10, args.steps // 10), max(3, args.warmup // 10), barrier)}}))
        return

    gm = synthetic_genetic_map(N_MARKERS, n_chr=10, n_traits=1, seed=1)
    geno = synthetic_genotypes(N_GERMPLASM, N_MARKERS, seed=0)

    def make_env(env_rng, **kw):
        return VecBreedGym(
            num_envs=N_ENVS, initial_population=geno, individual_per_gen=N_PER_ENV, num_generations=N_GENERATIONS,
            autoreset=True, reward_shaping=True, env_rng=env_rng, sync_rewards=False, genetic_map=gm, **kw,
        )

    # Headline: the drop-in configuration, all defaults.  env_rng="shared" is what the reference
    # computes -- its jax.vmap over an un-batched simulator key hands every env the same gamete keys
    # (vec_env.py:75-77) -- so per step the Threefry work is n x 2 gametes, applied to E x n crosses.
    # storage="bits" (default): the populations live as bit planes between steps; every step still
    # writes and returns the byte-layout observation (observations="eager").
    env = make_env("shared")
    # N ranks = one vector env of N x 16 envs, 16 per rank: the same seed everywhere, every rank draws its envs'
    # founders through its env window (world-size invariant; still no collective on the data path)
    env.simulator.engine.set_env_window(rank * N_ENVS, world * N_ENVS)
    dev = env.device
    arng = np.random.default_rng(2 + rank)
    actions = [
        torch.from_numpy(arng.integers(0, N_PER_ENV, (N_ENVS, N_PER_ENV, 2), dtype=np.int32)).to(dev)
        for _ in range(16)
    ]

    env.reset(seed=100)
    dt, t_issue = time_steps(env, actions, args.steps, args.warmup, barrier)
    t = all_reduce(torch.tensor([dt], dtype=torch.float64, device=dev), dist.ReduceOp.MAX)
    dt_max = float(t.item())
    distributed = what_the_ranks_saw(dt)
    # a long sample behind the driver's: the same loop for --long-steps further steps in the same process, so that a
    # reader can see the short sample (the driver's --steps 20 is 0.5 ms) is not luck.  `value` stays on --steps.
    dt_long = None
    if args.long_steps > 0:
        dt_l, _ = time_steps(env, actions, args.long_steps, 0, barrier)
        dt_long = float(all_reduce(torch.tensor([dt_l], dtype=torch.float64, device=dev), dist.ReduceOp.MAX).item())
    crosses_per_step = N_ENVS * N_PER_ENV
    kern_ms = time_cross_launches(env, actions, max(200, args.steps))

    # The same steps with an independent Threefry stream per cross (an extension: not what the
    # reference computes).  VALU-bound: 2 x 10k x 36 integer ops per cross.
    ti = kern_ms_i = None
    if not args.headline_only:
        env_i = make_env("independent")
        env_i.simulator.engine.set_env_window(rank * N_ENVS, world * N_ENVS)
        env_i.reset(seed=100)
        dt_i, _ = time_steps(env_i, actions, args.steps, args.warmup, barrier)
        ti = all_reduce(torch.tensor([dt_i], dtype=torch.float64, device=dev), dist.ReduceOp.MAX)
        kern_ms_i = time_cross_launches(env_i, actions, max(100, args.steps))
        del env_i

    # The other storage / observation choices of the same env, same outputs (one GPU only: extra
    # lines, not the headline): byte-layout state, and bit planes with on-demand observations.
    variants = {}
    if world == 1 and not args.headline_only:
        for name, kw in (("byte_layout_storage", dict(storage="bytes")),
                         ("lazy_observations", dict(storage="bits", observations="lazy"))):
            env_v = make_env("shared", **kw)
            env_v.reset(seed=100)
            dt_v, _ = time_steps(env_v, actions, args.steps, args.warmup, barrier)
            kms = time_cross_launches(env_v, actions, max(100, args.steps))
            nb = cross_bytes(N_MARKERS, crosses_per_step, N_PER_ENV, kw["storage"], kw.get("observations", "eager"))
            # requested bytes over time, NOT an HBM fraction: at this size the parents of a generation are the
            # previous launch's output and come out of the Infinity Cache (the byte layout asks for 246 MB)
            variants[name] = {"value": args.steps * crosses_per_step / dt_v, "ms_per_step": dt_v / args.steps * 1e3,
                              "kernel_ms": kms, "bytes_requested_per_launch": nb,
                              "requested_GBps": nb / (kms * 1e-3) / 1e9}
            del env_v

    # bytes the headline launch must move in the layout the env uses (bit planes + eager observation):
    # 2.75 M per cross + the mask rows; the reference-layout figure of SURVEY 8d (6 M per cross: two
    # 2M-byte parents read, 2M written) is reported beside it as what an API-level observer would count.
    nbytes = cross_bytes(N_MARKERS, crosses_per_step, N_PER_ENV, "bits", "eager")
    traffic, traffic_source = committed_traffic("packed_cross_kernel<1,4,2,false,")  # (the instance without riders, either layout: DESIGN 4f)

    if rank == 0:
        line = {
            "metric": "crosses/sec (vector env.step) at 370 pop x 10k markers",
            "value": world * args.steps * crosses_per_step / dt_max,
            "unit": "crosses/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[1]: 370x10k germplasm, 16 envs x 256 = 4096 crosses/gen, GEBV reward "
                            "every step, 10-generation episodes with autoreset",
                "n_markers": N_MARKERS, "crosses_per_step_per_gpu": crosses_per_step,
                "env_rng": "shared (reference semantics: vmap over an un-batched key, vec_env.py:75-77)",
                "storage": "bits (default): 1 bit per allele between steps, byte-layout observation written every step",
                "sharding": "envs across ranks, no data-path collective",
            },
            "roofline": roofline_of(
                "packed_cross_kernel<TT=1,MW=4,R=2> (row role: bit-plane parents -> select -> byte-layout observation + "
                "bit-plane state + GEBV + reward; mask role: next generation's Threefry)",
                nbytes, kern_ms, traffic=traffic, traffic_source=traffic_source,
                bytes_model="per cross: 2 x M/4 parents read + M/4 state written + 2M observation written; per keyed "
                            "cross: M/4 mask row read + M/4 written for the next generation",
                api_level_GBps=6 * N_MARKERS * crosses_per_step / (kern_ms * 1e-3) / 1e9,
                api_level_note="SURVEY 8d's reference-layout count (6M per cross) over the same time: what the byte layout "
                               "would have had to move; not a roofline fraction of this kernel",
                kernel_crosses_per_s=crosses_per_step / (kern_ms * 1e-3),
                step_frac_of_hbm_peak=nbytes / (dt_max / args.steps) / 1e9 / HBM_PEAK_GBPS,
            ),
            **({"value_long": world * args.long_steps * crosses_per_step / dt_long, "steps_long": args.long_steps,
                "ms_per_step_long": dt_long / args.long_steps * 1e3} if dt_long else {}),
            **({"distributed": distributed} if distributed else {}),
            "host_issue_ms_per_step": t_issue / args.steps * 1e3,
            "settle_s": SETTLE_S,  # untimed steps for that long before the W warm-up steps (clock settling)
        }
        if ti is not None:
            line["independent_rng"] = {
                "note": "one Threefry stream per cross (env_rng='independent', an extension); VALU-bound",
                "value": world * args.steps * crosses_per_step / float(ti.item()),
                "ms_per_step": float(ti.item()) / args.steps * 1e3,
                "kernel_ms": kern_ms_i,
            }
        if variants:
            variants["byte_layout_storage"]["note"] = "storage='bytes': the byte layout is the state; cross_kernel<1,4,2>"
            variants["lazy_observations"]["note"] = ("storage='bits', observations='lazy': the step hands out a handle "
                                                      "that expands to the byte layout on demand")
            line.update(variants)
        if world == 1 and not args.headline_only:
            del env
            torch.cuda.empty_cache()
            line["configs"] = {"c3": bench_c3(max(10, args.steps // 10), max(3, args.warmup // 10), barrier),
                               "c5": bench_c5(max(20, args.steps // 4)),
                               "c4_one_rank": bench_c4_one_rank(max(10, args.steps // 10), max(3, args.warmup // 10), barrier)}
        if world == 1 and not args.no_cpu_baseline and not args.headline_only:
            line["cpu_baseline"] = cpu_baseline(gm, geno, 0)
            line["speedup_vs_cpu_baseline"] = line["value"] / line["cpu_baseline"]["value"]
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

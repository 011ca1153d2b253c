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

The JSON line also carries, at N = 1: ``roofline`` (dominant kernel of the headline, named as rocprofv3 prints it
(KERNELS): ``frac`` = bytes it must move for the storage in use / its HIP-event time as a step issues it / 8 TB/s;
``frac_from_profiles`` = the same bytes / the committed rocprofv3 average of that kernel (profiles/<ROUND>_kernel_stats_*);
``traffic`` = memory-side bytes per launch from PMC counters, measured by this run in child passes under
``rocprofv3 --pmc`` with the kernel's FETCH_SIZE factor (FETCH_FACTORS); ``bound`` = the roofline priced against,
``binds`` = what really limits the launch), ``configs`` (configs[2], configs[3] as one rank sees it and configs[4], the
same way), ``cpu_baseline`` (the C oracle on the host's cores; the NumPy restatement and one C thread beside it).

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


def time_cross_launches(env, actions, reps, with_gebv=True, own_path=True, with_reward=False):
    """The hot-path launch alone (the fused two-role launch: streams this generation, makes the next
    one's masks), in the env's own storage layout, through the same C entry point the env uses but
    without the reset.  ``with_reward=True`` (the env's own path only): exactly the launch a step of an episode
    issues -- the per-env reward noted inside the launch, the next step's reward block primed by it.
    ``with_gebv=False`` leaves the GEBV out: where it is a pass of its own (long rows, three traits and
    more) that isolates the cross launch."""
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
            slot, _ = ring.step(cur[0].data_ptr(), n, actions[i % len(actions)], key, flags, with_reward, with_reward)
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


ROUND = "r05"  # the round whose rocprofv3 summaries under profiles/ the line is checked against (frac_from_profiles)

# Every kernel a roofline object of the line names, exactly as rocprofv3 prints it (the demangled symbol without the
# "void bh::" prefix and the argument list) -> the profile of that configuration under profiles/<ROUND>_kernel_stats_*.
# tests/test_isa.py checks that every one of them is a symbol of the library's emitted gfx950 ISA.
KERNELS = {
    "headline": ("packed_cross_kernel<1, 4, 2, false, 0>", "c2"),
    "headline_horizon": ("packed_cross_kernel<1, 4, 2, true, 0>", "c2"),
    "c3_cross": ("packed_long_kernel<1, 8>", "c3"),
    "c3_finish": ("mask_finish4_kernel", "c3"),
    "c3_gebv": ("gebv_bits_kernel<1, 4>", "c3"),
    "c3_reward": ("reward_max_kernel", "c3"),
    "c5_bytes": ("gebv_bytes_kernel<2>", "c5"),
    "c5_bits": ("gebv_bits_regs_kernel<2>", "c5"),
}


def norm_kernel(name):
    """rocprofv3's kernel name -> the form of KERNELS: no return type, namespace, argument list or blanks."""
    return name.split("(")[0].replace("void ", "").replace("bh::", "").replace(" ", "")


def profile_avg_us(which, cfg=None):
    """Average duration (us) and call count of kernel KERNELS[which] in the committed rocprofv3 --kernel-trace --stats
    summary of its configuration (profiles/<ROUND>_kernel_stats_<cfg>.csv), or (None, None, file) when there is none."""
    import csv

    name, cfg0 = KERNELS[which]
    cfg = cfg or cfg0
    f = ROOT / "profiles" / f"{ROUND}_kernel_stats_{cfg}.csv"
    if f.exists():
        for row in csv.DictReader(open(f)):
            if norm_kernel(row["Name"]) == norm_kernel(name):
                return float(row["AverageNs"]) / 1e3, int(row["Calls"]), f"profiles/{f.name}"
    return None, None, f"profiles/{f.name}"


def roofline_of(which, note, nbytes, ms, binds, extra_kernels=(), profile_cfg=None, **extra):
    """The roofline object of kernel KERNELS[which] (+ `extra_kernels` when the timed launch is a pair): `bound` is the
    roofline the fraction is PRICED against (the contract's enum: HBM for every kernel of this path -- SURVEY 8d),
    `binds` says what really limits the launch where that is something else."""
    achieved = nbytes / (ms * 1e-3) / 1e9
    names = [KERNELS[w][0] for w in (which, *extra_kernels)]
    if extra.get("traffic") is not None:
        # what the counters are (MI355X_MICROARCH.md, HBM section): requests at the memory side of the L2, Infinity-Cache
        # hits included -- an upper bound on HBM bytes, not HBM bytes
        extra.setdefault("traffic_is", "memory-side (fabric) bytes per launch: fetch_factor x FETCH_SIZE + WRITE_SIZE (KiB); "
                                       "Infinity-Cache hits are counted, so at cache-resident sizes this is an upper bound on HBM bytes")
    r = {"kernel": " + ".join(names), "kernel_note": note, "bound": "hbm", "binds": binds, "achieved": achieved,
         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "bytes_per_launch": nbytes, "kernel_ms": ms,
         "kernel_ms_is": "HIP events on the launch stream around back-to-back launches of exactly what a step issues, "
                         "clocks settled (time_events)"}
    # the same bytes over the committed rocprofv3 average of the same kernel(s): what a reader recomputes from profiles/
    us = [profile_avg_us(w, profile_cfg) for w in (which, *extra_kernels)]
    if all(u[0] is not None for u in us):
        tot = sum(u[0] for u in us)
        r["frac_from_profiles"] = nbytes / (tot * 1e-6) / 1e9 / HBM_PEAK_GBPS
        r["profiles_avg_us"] = tot
        r["profiles_source"] = ", ".join(f"{u[2]}: {KERNELS[w][0]} avg {u[0]:.2f} us over {u[1]} launches"
                                         for u, w in zip(us, (which, *extra_kernels)))
    else:
        r["frac_from_profiles"] = None
        r["profiles_source"] = f"no {ROUND} summary committed yet for " + ", ".join(n for n, u in zip(names, us) if u[0] is None)
    r.update(extra)
    return r


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
                                                (2048, 100_000, 4, 2048, 21, 8, True)):  # (+ the step's GEBV pass over the offspring)
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
            name = norm_kernel(k)
            _LIVE_TRAFFIC[name] = ((fetch_factor(name) * f + w) * 1024.0, nf, f, w)
        _LIVE_NOTE = "ok"
    except Exception as e:  # a missing counter, a timeout: the committed passes remain
        _LIVE_NOTE = f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def fetch_factor(kernel_name):
    """What FETCH_SIZE has to be multiplied by.  The guide's gfx950 correction (x 2) is for wide coalesced streaming
    reads (16 bytes per lane, consecutive lanes consecutive addresses), which the counter tallies at half their bytes.
    Calibrated per kernel where its reads are of another shape (profiles/r05_pmc_calibration.txt):
      gebv_bits_regs_kernel<2> operands by global_load_dwordx4 of consecutive lanes: x 1.92 measured with the row reads
                               left out (3.33 MB raw for 6.4 MB); rows by global_load_dwordx4, eight lanes per 128-byte
                               line of each of eight rows per instruction.  While a 100k-marker row was 25,008 bytes and
                               every such piece straddled two lines the rows were tallied at 58 % (x 1.71, blend 1.73);
                               since long rows start on lines (bh_packed_pitch, second session of round 5) the raw count of
                               the shipped kernel is 30.06 MB for 51.4 MB of rows + 6.4 MB of operands that cannot come
                               from an L2: at least x 1.92 for both kinds -- whole aligned lines are tallied at half, as the
                               guide says                                                                          -> 1.92
      gebv_bits_kernel<1, 4>   rows and operands by LDS-DMA, eight lanes taking the eight 16-byte pieces of a line in a
                               permuted order.  Straddling two lines each (round 4) they were tallied in full (59.6 MB raw
                               for 51.2 MB); on lines the step's pass reads 113.7 MB raw for 205.5 MB of rows no cache can
                               hold + 3.2 MB of operands: at least x 1.84                              -> the guide's x 2
    Every other kernel of the line reads 16 bytes per lane from consecutive addresses (or 2-byte words of consecutive
    lanes, small next to its stores): the guide's x 2."""
    n = norm_kernel(kernel_name)
    for prefix, ff in FETCH_FACTORS.items():
        if n.startswith(norm_kernel(prefix)):
            return ff
    return 2.0


FETCH_FACTORS = {"gebv_bits_regs_kernel<": 1.92}


def committed_traffic(which):
    """PMC traffic (fetch_factor x FETCH_SIZE + WRITE_SIZE, per launch) of kernel KERNELS[which]: measured by this run
    (live_traffic) where that worked, else from the committed rocprofv3 passes of THIS round under profiles/ -- a
    measurement of an earlier run of the same command, NOT of this run.  The kernel is matched by its exact name; a
    kernel no committed pass knows has no traffic figure (never another kernel's, never another round's under the same
    label).  The source is named either way."""
    want = norm_kernel(KERNELS[which][0])
    if want in _LIVE_TRAFFIC:
        b, n, f, w = _LIVE_TRAFFIC[want]
        return b, (f"measured by this run: child passes under rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE "
                   f"(separate), mean of {n} launches of {KERNELS[which][0]}: {fetch_factor(want):g} x FETCH_SIZE ({f:.0f} KiB) "
                   f"+ WRITE_SIZE ({w:.0f} KiB)")
    f = ROOT / "profiles" / f"{ROUND}_pmc_traffic.json"
    why = _LIVE_NOTE or "live passes not requested"
    if f.exists():
        try:
            entries = json.loads(f.read_text()).get("kernels", [])
        except Exception:
            entries = []
        for e in entries:
            if norm_kernel(str(e.get("kernel", ""))) == want:
                return e.get("hbm_bytes_per_launch"), (f"profiles/{f.name} (rocprofv3 --pmc passes of an earlier run of the same "
                                                       f"command; not measured by this run: {why})")
    return None, f"none: not measured by this run ({why}) and no committed {ROUND} PMC pass names {KERNELS[which][0]}"


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
        # the launches a step issues, on the env's own blocks in its own alternation (StepRing.step, as the headline's):
        # the long-row launch, the finish, the GEBV pass, the reward reduction -- without the resets
        kms_step = time_cross_launches(env, actions, max(10, steps), with_reward=True)
        nbytes = cross_bytes(M, E * n, n, "bits", obs)
        site_bytes = 2 * 4 * n * (M // 128 + 2) * 8  # site words written by the launch and read back by the finish kernel
        gebv_bytes = E * n * packed_pitch(M) + M * 64   # the offspring's bit planes read again + the digit operands
        seq = ("c3_cross", "c3_finish", "c3_gebv", "c3_reward")
        traffic = [committed_traffic(w) for w in seq[:3]]
        eager = obs == "eager"  # (the PMC passes are of the eager configuration)
        cfg = "c3" if eager else "c3lazy"
        us_gebv = profile_avg_us("c3_gebv", cfg)
        r = {"value": steps * E * n / dt, "unit": "crosses/s", "ms_per_step": dt / steps * 1e3,
             "roofline": roofline_of(
                 "c3_cross", "the launches of a step: the long-row launch -- four row waves stream rows without GEBV (LDS-DMA "
                 "ring), eight mask waves draw the next generation's sites one gamete stream per lane: 2048 keyed crosses x 100k "
                 "markers of Threefry --, site words -> masks (the finish kernel), the matrix-core GEBV pass over the offspring, "
                 "the per-env reward reduction", nbytes + site_bytes + gebv_bytes, kms_step,
                 binds="the long-row launch (0.47-0.50 of the ~0.56 ms): its row role runs at 0.94 of the achievable ~6.3 TB/s when "
                       "alone (415 us) and the launch as a whole at 83 % of its integer issue (289.8 M VALU wave-instructions, most "
                       "of them JAX's bit stream): two floors 0.1 ms apart; then ~0.09 ms of tail (finish 0.029, GEBV pass 0.055, reward 0.005) "
                       "that runs behind the launch's 1.8 GB of stores",
                 extra_kernels=seq[1:], profile_cfg=cfg,
                 traffic=sum(t[0] for t in traffic) if eager and all(t[0] for t in traffic) else None,
                 traffic_source=traffic[0][1] if eager else None,
                 api_level_GBps=6 * M * E * n / (kms_step * 1e-3) / 1e9),
             "gebv_pass": {"kernel": KERNELS["c3_gebv"][0], "bytes_per_launch": gebv_bytes, "profiles_avg_us": us_gebv[0],
                           "frac_from_profiles": gebv_bytes / (us_gebv[0] * 1e-6) / 1e9 / HBM_PEAK_GBPS if us_gebv[0] else None,
                           "profiles_source": us_gebv[2],
                           "note": "the step's GEBV: a matrix-core pass over the bit-plane offspring the launch just wrote; timed by "
                                   "rocprofv3 only (inside the step it cannot be told apart by events); its rows, ~10 us of ramp / "
                                   "first operands / way out around a stream at ~5 TB/s"},
             "step_frac_of_hbm_peak": (nbytes + site_bytes + gebv_bytes) / (dt / steps) / 1e9 / HBM_PEAK_GBPS}
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
                      "roofline": roofline_of(
                          "c5_bytes", "bh_gebv, byte-layout rows in one pass: bit planes built in LDS, digit GEMMs on the matrix "
                          "cores, 64-bit sums by atomics", N * 2 * M + eff_bytes, ms_bytes, binds="hbm (its rows: 410 MB)",
                          traffic=committed_traffic("c5_bytes")[0], traffic_source=committed_traffic("c5_bytes")[1])},
        "bit_planes": {"ms": ms_bits, "individuals_per_s": N / (ms_bits * 1e-3),
                       "roofline": roofline_of(
                           "c5_bits", "bh_gather_packed, bit-plane rows, two column tiles: the register-staged pass -- rows as whole "
                           "128-byte lines turned in LDS, digit operands through registers into two LDS buffers, a barrier per "
                           "128 markers, cells completed by 64-bit atomics that carry the arrival count",
                           N * packed_pitch(M) + eff_bytes, ms_bits,
                           binds="neither bytes nor MFMA: ramp, first operands and the way out of the K slices are ~14 of its ~23 us; "
                                 "the stream between runs at ~55 % of the matrix cores' rate with VALU (bits -> int8 operands), LDS "
                                 "and MFMA each busy more than half of it (profiles/r04_ab_gebv.txt)",
                           traffic=committed_traffic("c5_bits")[0], traffic_source=committed_traffic("c5_bits")[1],
                           fetch_factor=fetch_factor(KERNELS["c5_bits"][0]),
                           mfma_int8_Tops=2.0 * N * M * 64 / (ms_bits * 1e-3) / 1e12, mfma_int8_peak_Tops=5000.0),
                       "api_level_GBps": N * 2 * M / (ms_bits * 1e-3) / 1e9},
    }


def population_generation(steps, warmup, rank, world, local_rank, use_dist, barrier, all_reduce=None, seen=None, n_local=8192):
    """BASELINE configs[3]: one population sharded row-wise, 8192 x 100k per rank, k = n_total // 10.  Returns the
    JSON line of the population mode (rank 0; None elsewhere).  `n_local`: rows per rank (65,536 with one rank = the
    whole population of configs[3] on one device)."""
    import torch
    import torch.distributed as dist

    from breedgym_amd.simulator import Simulator
    from breedgym_amd.synthetic import synthetic_genetic_map, synthetic_genotypes
    from breedgym_amd.vector.distributed import ShardedPopulation

    M = 100_000
    n_total = n_local * world
    k = n_total // 10
    gm = synthetic_genetic_map(M, n_chr=21, n_traits=1, seed=1)
    sim = Simulator(genetic_map=gm, seed=5, device=local_rank)
    eng = sim.engine
    base = synthetic_genotypes(512, M, seed=10 + rank)  # a rank's shard: 8192 rows drawn from 512 distinct ones
    rows = eng.upload_population(base)
    pick = torch.from_numpy(np.random.default_rng(rank).integers(0, 512, n_local, dtype=np.int32)).to(eng.device)
    shard = eng.empty_population(n_local)  # (byte rows: 13.1 GB where one rank holds all 65,536)
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
        "metric": f"crosses/sec (population-sharded generation: top-k GEBV selection + crosses) at {n_local} x 100k markers per GPU",
        "value": world * steps * n_local / dt_max, "unit": "crosses/s", "n_gpus": world, "steps": steps,
        "warmup": warmup, "ms_per_step": dt_max / steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {
            "workload": f"BASELINE configs[3]: {n_total} individuals x 100k markers over {world} GPU(s), {n_local} per rank, "
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
    k = n // 10: of the 21.5 M diallel pairs only the 8,192 wanted positions are followed backwards through JAX's three
    shuffle rounds -- csrc/shuffle_sparse.h; nothing is sorted)."""
    import torch

    from breedgym_amd import HipEngine

    line = population_generation(steps, warmup, 0, 1, torch.cuda.current_device(), False, barrier)
    out = {"workload": line["config"]["workload"], "ms_per_generation": line["ms_per_step"], "value": line["value"],
           "unit": line["unit"], "bytes_per_generation": line["bytes_per_generation_per_gpu"], "bound": line["bound"],
           "note": "ONE rank's share, measured on one GPU; the 8-rank run (bench.py --gpus 8 --mode population) adds the two "
                   "all-gathers over RCCL, which no 1-GPU run can time"}
    # ... and the WHOLE population of configs[3] on this one device (1.6 GB of bit planes; the byte rows it is loaded from
    # 13.1 GB): the generation the eight ranks compute between them -- GEBV of 65,536, top-6,553, the winners' pool, 65,536
    # crosses of one flat key split -- with no exchange step, because nothing has to be exchanged
    torch.cuda.empty_cache()
    w = population_generation(max(5, steps // 2), 2, 0, 1, torch.cuda.current_device(), False, barrier, n_local=65_536)
    out["whole_population_on_one_gpu"] = {
        "workload": w["config"]["workload"], "ms_per_generation": w["ms_per_step"], "value": w["value"], "unit": w["unit"],
        "generations": w["steps"], "bound": w["bound"],
        "note": "all 65,536 individuals x 100k markers on ONE MI355X, world_size = 1: what configs[3] asks eight GPUs for fits "
                "one many times over as bit planes; per generation 8 x one rank's Threefry work and no collective",
    }
    torch.cuda.empty_cache()
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


def count_gpus():
    """GPUs this box offers, counted by a short-lived CHILD process (torch.cuda.device_count() goes through
    hipGetDeviceCount -> hipInit on ROCm; the child has exited before any rank starts): the launcher itself never loads
    the HIP runtime."""
    import subprocess

    try:
        r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True,
                           text=True, timeout=300)
        return int(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else 0
    except Exception:
        return 0


def self_launch(args):
    """``python bench.py --gpus N`` without a launcher: start the N ranks as fresh child processes (this process has
    not touched the GPU and never will), wait, relay rank 0's one JSON line.  -> exit code."""
    import socket
    import subprocess

    n = args.gpus
    ndev = count_gpus()  # by a child that has exited: the launcher never loads the HIP runtime
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
    import tempfile

    rank0_out = tempfile.TemporaryFile()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r % ndev), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BENCH_BACKEND=backend, BENCH_SELF_LAUNCHED="1",
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        # rank 0's stdout carries the line (to a file: nobody has to drain a pipe while the ranks are watched); the
        # other ranks' goes to stderr
        procs.append(subprocess.Popen(argv, env=env, stdout=rank0_out if r == 0 else sys.stderr.fileno()))
    # watch ALL ranks: one that dies (a fault, a kill, an exception) leaves the others waiting in a collective for as
    # long as the backend's timeout -- so the first non-zero exit ends the job: the survivors get a few seconds, then
    # they are ended by their own pids, and the launcher exits non-zero
    codes = [None] * n
    failed_at = None
    while any(c is None for c in codes):
        for r, p in enumerate(procs):
            if codes[r] is None:
                codes[r] = p.poll()
        if failed_at is None and any(c not in (None, 0) for c in codes):
            failed_at = time.monotonic()
        if failed_at is not None and time.monotonic() - failed_at > 10.0:
            for r, p in enumerate(procs):
                if codes[r] is None:
                    p.kill()
                    codes[r] = p.wait()
        time.sleep(0.05)
    rank0_out.seek(0)
    out = rank0_out.read()
    rank0_out.close()
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

    if os.environ.get("BENCH_TEST_KILL_RANK") == str(rank) and use_dist:
        # tests/test_gpu_distributed.py: a rank that dies once the group is up (SIGKILL to its own pid) -- the launcher
        # must end the job with a non-zero exit instead of waiting for the others' collective to time out
        import signal

        os.kill(os.getpid(), signal.SIGKILL)

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
    # the launch a step issues (reward noted in the launch, the next reward block primed): what `roofline` prices; and
    # the same launch without the reward, as rounds 1-4 timed it
    kern_ms = time_cross_launches(env, actions, max(200, args.steps), with_reward=True)
    kern_ms_noreward = time_cross_launches(env, actions, max(200, args.steps))

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
    traffic, traffic_source = committed_traffic("headline")  # (the instance without riders: nine of an episode's ten launches)

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
                "headline", "row role: bit-plane parents -> select -> byte-layout observation + bit-plane state + GEBV + per-env "
                "reward; mask role: next generation's Threefry.  Nine of an episode's ten launches; the horizon step's is " +
                KERNELS["headline_horizon"][0] + " (no observation, + the next episode's founder gather)",
                nbytes, kern_ms,
                binds="not HBM: the working set (two recycled 82 MB observation blocks + 10 MB of state) lives in the 256 MB "
                      "Infinity Cache and the launch is byte-insensitive (without the 82 MB observation: lazy_observations.kernel_ms). "
                      "What binds is the chip's store path -- a kernel of this grid that does nothing but these stores takes 14.0 us "
                      "per back-to-back launch (6.6 TB/s) and 16.6 us with the ~3 us this launch needs before its first store "
                      "(entry barrier, indices, first words): profiles/r05_store_ceiling.txt -- and that fixed head; the mask role "
                      "beside the row streams costs 1.8 us, the way out ~1.1; the HBM fraction is kept as the stated convention",
                traffic=traffic, traffic_source=traffic_source, kernel_ms_without_reward=kern_ms_noreward,
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

#!/usr/bin/env python3
"""bench.py — SPH throughput of the MI355X stepper on the BASELINE.json workloads.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg1|cfg0|cfg4] [--no-cpu]

A "step" is one pass of the hot path (kick/drift, neighbour structure, density/EOS, force/kick:
pi_sph_fluid.c:612-641) over every fluid particle of the scene.  The neighbour structure (counting
sort + neighbour lists) is rebuilt when a particle has moved more than half the skin, as decided on
the device every step; `neighbour_rebuilds_per_step` reports how often that was in the timed run.  Inputs are resident in HBM when
the timed region starts (sph_create has run); the timed region is exactly K steps, bracketed by a
device synchronisation (and a barrier across ranks when N > 1), MAX over ranks.

Workload at N = 1: cfg2, the 2 000 000-particle dam break (4000 x 500 block, box 1200 x 60 m), the
configuration BASELINE.json's north_star quotes its single-GPU target and roofline on.  At N > 1 the
same scene is extended to N slabs of 2M particles (cfg3 = the N = 4 member of that family), i.e.
weak scaling; `value` is the whole-job Mparticle-steps/s.  cfg1 (the 262 144-particle drop) is also
measured at N = 1 and reported under "also".

One JSON line on stdout (rank 0).  Everything the oracle touches is the cpu_baseline leg only.
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md "HBM3E peak BW"); ~6290 GB/s measured copy
VALU_CYCLES, TRANS_CYCLES, N_SIMD, CLOCK_GHZ = 2.67, 8.75, 1024, 2.4      # tools/ubench_valu on MI355X; 256 CUs x 4 SIMDs
TRANS_SHARE = {"force_kick": 2.0 / 29.0, "density_eos": 1.0 / 14.0}      # transcendental instructions among a walker's VALU instructions (from the ISA)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


# The contract is ONE JSON line on stdout.  Libraries print there too (RCCL writes a version banner with printf when
# its first communicator is created), so the process's stdout is pointed at stderr for the whole run and the result
# line goes to the original descriptor.
_REAL_STDOUT = None


def quiet_stdout():
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit(obj):
    line = (json.dumps(obj) + "\n").encode()
    sys.stdout.flush()
    os.write(_REAL_STDOUT if _REAL_STDOUT is not None else 1, line)


def box_calibration(sph):
    """What THIS box delivers (sph_box_calibrate, include/sph_diag.h): 3 x ~50 ms of a streaming copy over 1 GiB and of a saturated
    v_fma_f32 stream, medians.  The boxes of the pool differ by +-5-8 % for one build of the library; with this record a figure can be
    normalised for its box (`*_vs_copy`: fractions of the MEASURED copy bandwidth instead of the 8 TB/s specification)."""
    if os.environ.get("SPH_BENCH_BOX", "1") == "0":      # (profiles/collect.sh: the calibration kernels would fill the kernel trace)
        return {"status": "skipped ($SPH_BENCH_BOX=0)"}
    try:
        b = sph.box_calibrate(0, 3)
        b["what"] = ("copy_gbs: read + write bytes / time of a float4 copy kernel over 1 GiB; valu_cycles: SIMD-cycles per wave-instruction "
                     "of an independent v_fma_f32 stream on every SIMD, priced at the nominal %.1f GHz (tools/ubench_valu); clock_ghz: "
                     "s_memtime ticks per s_memrealtime tick x 100 MHz under that load; medians of %d runs of ~50 ms each, before the timed region" % (CLOCK_GHZ, b["repeats"]))
        return b
    except Exception as e_:      # (reported, never raised)
        log("box calibration failed: %r" % (e_,))
        return {"status": "failed: %r" % (e_,)}


def vs_copy(rf, box):
    """the roofline's fractions once more against the copy bandwidth measured on this box"""
    c = (box or {}).get("copy_gbs")
    if not c or not rf:
        return rf
    rf = dict(rf)
    if rf.get("achieved"):
        rf["frac_vs_copy"] = round(rf["achieved"] / c, 4)
    if rf.get("step_achieved"):
        rf["step_frac_vs_copy"] = round(rf["step_achieved"] / c, 4)
    if rf.get("step_frac_executed") is not None:
        rf["step_frac_executed_vs_copy"] = round(rf["step_frac_executed"] * HBM_PEAK_GBS / c, 4)
    return rf


# launches per kernel of the duration measurement in front of the timed region (run_single): enough of them that a fresh
# box has reached its running clocks when the window starts — the driver times 20 steps (2 ms) after 5, and a GPU that
# has worked for 5 ms in its life runs the force pass in 64 us instead of 53
PRE_REPS = int(os.environ.get("SPH_BENCH_PRE_REPS", "300"))


def run_single(sph, name, steps, warmup, profile_steps=20, skin=None, tilt=False, windows=1, load_state=None, save_state=None):
    """`windows` timed windows of `steps` steps each of one scene on device 0, after `warmup` steps; returns a result dict
    (rates: the median window; windows = 1: exactly `steps` timed steps).  tilt: gravity from the scripted tilt trace
    (sph_gravity: 15 deg, 8 s period, re-sampled every 0.1 s of simulated time like the reference's 10 Hz poll,
    pi_sph_fluid.c:455-461), one sph_step call per step as the reference re-reads g every step (:632)."""
    prm, f, b = sph.scene(name) if name != "cfg2" else sph.dam_break(1)
    if skin is not None:
        prm.skin = prm.skin_min = skin      # a fixed skin
    elif os.environ.get("SPH_BENCH_SKIN_MIN") or os.environ.get("SPH_BENCH_SKIN_MAX"):      # (sweeps: the ends of the adaptive range)
        prm.skin_min = float(os.environ.get("SPH_BENCH_SKIN_MIN", prm.skin_min))
        prm.skin = float(os.environ.get("SPH_BENCH_SKIN_MAX", prm.skin))
    n = len(f)
    grav = sph.GravitySource(sph.GRAVITY_TILT, 9.81) if tilt else None
    dt_sim = float(np.float32(prm.dt))
    sim = [0]

    def advance(k):
        if grav is None:
            ctx.step(k, 0.0, -9.81)
            return
        for _ in range(k):
            gx, gy = grav.sample(sim[0] * dt_sim)
            ctx.step(1, gx, gy)
            sim[0] += 1

    t0 = time.time()
    g0 = grav.sample(0.0) if grav else (0.0, -9.81)
    ctx = sph.Context(prm, f, b, g0[0], g0[1], device=0)
    create_s = time.time() - t0
    del f
    if load_state:          # a checkpoint (sph_read_particles + sph_read_accel of an earlier run): profiling a developed flow
        ck = np.load(load_state)      # without stepping up to it under the profiler
        ctx.upload_state(ck["particles"].view(sph.PARTICLE).reshape(-1))
        ctx.upload_accel(ck["du"], ck["dv"])
    advance(warmup)
    ctx.sync()
    if save_state:
        du_, dv_ = ctx.read_accel()
        np.savez(save_state, particles=ctx.read_particles(), du=du_, dv=dv_)
    # the two heavy kernels at the START of the timed region (back-to-back launches of the idempotent kernels on the live
    # state); again at its end below: their mean is the figure for "the kernel's average duration over the timed region"
    def spec_ms(reps):      # the density launch as the step issues it (speculative, with the criterion's jobs) + its reset launch
        try:
            return ctx.time_kernel("density_spec", reps)
        except sph.SphError:
            return None
    k_begin = (ctx.time_kernel("density_eos", PRE_REPS), ctx.time_kernel("force_kick", PRE_REPS), spec_ms(50)) if warmup > 0 and not load_state else None
    rates, rebuilt = [], []
    v_begin, why_begin = ctx.verify_stats(), ctx.rebuild_reasons()
    try:
        rep_begin = ctx.repair_stats()
    except AttributeError:      # (an older build of the library: A/B runs)
        rep_begin = (0, 0, 0, 0)
        ctx.repair_stats = lambda: (0, 0, 0, 0)
    for _ in range(windows):
        r0, _d = ctx.rebuild_stats()
        t0 = time.perf_counter()
        advance(steps)
        ctx.sync()
        dt = time.perf_counter() - t0
        r1, _d = ctx.rebuild_stats()
        rates.append(steps / dt)
        rebuilt.append((r1 - r0) / max(steps, 1))
    verified = (ctx.verify_stats() - v_begin) / max(steps * windows, 1)
    repairs = [a_ - b_ for a_, b_ in zip(ctx.repair_stats(), rep_begin)]
    why = [a_ - b_ for a_, b_ in zip(ctx.rebuild_reasons(), why_begin)]
    mid = int(np.argsort(rates)[len(rates) // 2])          # the median window (its own rebuild rate goes with it)
    kt = ctx.profile_steps(profile_steps, *(grav.sample(sim[0] * dt_sim) if grav else (0.0, -9.81)))      # HIP events on the kernels' own stream
    ctx.sync()
    # sph_profile_steps launches one kernel per phase of the rebuild chain with an event around each (sph_step itself
    # launches the chain as one kernel and eight steps as one graph): its sum is not the step time of the timed loop
    kt["profiled_step_separate_launches"] = kt.pop("step")
    # the two heavy kernels are idempotent: time back-to-back launches on the live state (no per-launch
    # event overhead; this is the figure that must agree with rocprofv3's average kernel duration)
    k_end = (ctx.time_kernel("density_eos", 50), ctx.time_kernel("force_kick", 50), spec_ms(50))
    if k_begin is None:
        k_begin = k_end
    kt["density_eos"] = 0.5 * (k_begin[0] + k_end[0])
    kt["force_kick"] = 0.5 * (k_begin[1] + k_end[1])
    # density_eos: the plain pass (the tiles alone).  What sph_step launches on this context is the SPECULATIVE pass — the same
    # tiles plus the check / verify jobs of the rebuild criterion as extra workgroups (rocprofv3: k_density_list<1, 0, true>) —
    # timed here back to back with the one-thread launch that resets the jobs' words between two of them (sph_diag.h)
    if k_begin[2] is not None and k_end[2] is not None:
        kt["density_spec_launch_plus_reset"] = 0.5 * (k_begin[2] + k_end[2])
    kt["density_eos_at_begin_end"] = [round(k_begin[0], 5), round(k_end[0], 5)]
    kt["force_kick_at_begin_end"] = [round(k_begin[1], 5), round(k_end[1], 5)]
    max_rho, max_speed = ctx.stats()
    rows, cols = ctx.grid_dims()
    rebuilds, direct_tiles = ctx.rebuild_stats()
    res = {"workload": name, "n_fluid": n, "n_boundary": len(b), "grid_cells": rows * cols,
           "steps_per_s": rates[mid], "ms_per_step": 1e3 / rates[mid],
           "mparticle_steps_per_s": rates[mid] * n / 1e6, "kernel_ms": kt,
           "max_rho": max_rho, "max_speed": max_speed, "create_s": create_s, "skin_frac": float(prm.skin),
           "skin_min_frac": float(min(prm.skin_min, prm.skin)), "skin_now": ctx.current_skin(),
           "verified_pairs_per_step": verified, "rebuild_requests": why, "list_repairs": repairs,
           "rebuilds": rebuilds, "direct_tiles": direct_tiles, "timed_rebuilds_per_step": rebuilt[mid],
           "window_steps_per_s": [round(x, 2) for x in rates], "window_rebuilds_per_step": [round(x, 4) for x in rebuilt],
           "windows": windows, "window_steps": steps, "warmup": warmup,
           "device_mb": ctx.device_bytes() / 1e6}
    ctx.close()
    return res


# SURVEY.md 8d, algorithmic bytes per particle and pass.  The 152 B step assumes the reference's structure: a rebuild of
# the neighbour structure every step (:626).  Here a rebuild happens only in some steps, so the passes that actually
# ran move fewer algorithmic bytes: P5 + P6 + the fused P1 (without its key) always, the sort passes per rebuild.
ALGO_ALWAYS = 16.6 + 40.0 + 40.0
ALGO_PER_REBUILD = 4.0 + 5.2 + 1.2 + 44.0
FORCE_FUSED_NET = 40.0 + 40.0 - 24.0        # the fusion keeps x, y, u, v, ax, ay in registers: P1's 24 B of reads never happen


def roofline(sph, res, traffic_key=None):
    """HBM roofline of the dominant kernel: algorithmic bytes per launch / its mean launch duration (HIP events on the
    kernels' own stream, back-to-back launches on the live state).  traffic: PMC bytes per launch of that kernel in the
    same regime of the same workload (profiles/traffic.json: collected offline with rocprofv3, profiles/README.md)."""
    kt = res["kernel_ms"]
    cand = {k: kt[k] for k in ("kick_drift", "density_eos", "force_kick")}
    dom = max(cand, key=cand.get)
    algo = sph.KERNEL_ALGO_BYTES[dom] * res["n_fluid"]
    achieved = algo / (kt[dom] * 1e-3) / 1e9
    step_bytes = sph.STEP_ALGO_BYTES * res["n_fluid"]
    step_gbs = step_bytes * res["steps_per_s"] / 1e9
    exec_bytes = (ALGO_ALWAYS + ALGO_PER_REBUILD * res["timed_rebuilds_per_step"]) * res["n_fluid"]
    exec_gbs = exec_bytes * res["steps_per_s"] / 1e9
    traffic, valu = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):          # PMC passes are collected offline with rocprofv3 (see profiles/README.md)
        try:
            entry = json.load(open(tpath)).get(traffic_key or res["workload"], {})
            traffic = entry.get(dom)
            v = entry.get("_valu", {}).get(dom)
            if v:      # the secondary ceiling: VALU issue.  SQ_INSTS_VALU wave-instructions per launch over 1024 SIMDs at the issue
                       # cost tools/ubench_valu measures on this part (2.67 cycles a plain f32 operation, 8.75 a transcendental one)
                if not v.get("trans"):      # (SQ_INSTS_VALU_TRANS reads 0 on this stack: the share from the kernels' code — the force pass
                    v = dict(v, trans=int(v["insts"] * TRANS_SHARE.get(dom, 0.0)))      # issues v_sqrt + v_rcp per ~29 instructions of a list entry, the density pass v_sqrt per ~14)
                cyc = ((v["insts"] - v["trans"]) * VALU_CYCLES + v["trans"] * TRANS_CYCLES) / N_SIMD
                busy_us = cyc / CLOCK_GHZ / 1e3
                valu = {"wave_insts_per_launch": v["insts"], "transcendental": v["trans"], "issue_us": round(busy_us, 2),
                        "frac_of_kernel": round(busy_us / (kt[dom] * 1e3), 4),
                        "model": "%.2f / %.2f SIMD-cycles per plain / transcendental wave-instruction (tools/ubench_valu), %d SIMDs, %.1f GHz"
                                 % (VALU_CYCLES, TRANS_CYCLES, N_SIMD, CLOCK_GHZ)}
        except Exception:
            traffic = None
    out = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "valu": valu,
           "algo_bytes_per_launch": algo, "kernel_ms": round(kt[dom], 5),
           "step_achieved": round(step_gbs, 1), "step_frac": round(step_gbs / HBM_PEAK_GBS, 4),
           "step_algo_bytes": step_bytes,
           # the same step priced on the passes that actually ran in the timed window (no credit for sort passes of
           # steps that did not rebuild)
           "step_executed_bytes": round(exec_bytes), "step_frac_executed": round(exec_gbs / HBM_PEAK_GBS, 4)}
    if dom == "force_kick":
        net = FORCE_FUSED_NET * res["n_fluid"]
        out["frac_fused_net"] = round(net / (kt[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)      # 56 B: what the fused kernel must move
    return out


_CPU_LEG = r"""
import importlib, json, os, sys, time
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "oracle"))
import numpy as np
import orc
sph = importlib.import_module("pi-sph-fluid_amd")
name, threads, nsteps, windows = {name!r}, {threads}, {nsteps}, {windows}
O = orc.Oracle("fast")
prm, f, b = sph.scene(name) if name != "cfg2" else sph.dam_break(1)
p = O.params((prm.x_min, prm.x_max, prm.y_min, prm.y_max))
of, ob = O.first_touch(f.view(orc.PARTICLE), threads), b.view(orc.PARTICLE).copy()      # pages touched by the threads that work on them
O.psi(p, ob)
du, dv = O.eval(p, of, ob, 0.0, -9.81, threads=threads)
du, dv = O.first_touch(du, threads), O.first_touch(dv, threads)
O.steps(p, of, ob, 0.0, -9.81, du, dv, 2, threads=threads)
rates = []
for w in range(windows):
    t0 = time.perf_counter()
    O.steps(p, of, ob, 0.0, -9.81, du, dv, nsteps, threads=threads)
    rates.append(nsteps / (time.perf_counter() - t0))
print(json.dumps({{"rates": rates, "n": len(f)}}))
"""


def physical_cores():
    """one logical CPU per physical core among those this process may run on"""
    allowed = sorted(os.sched_getaffinity(0))
    seen, cores = set(), []
    for cpu in allowed:
        try:
            sib = open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % cpu).read().strip()
        except OSError:
            sib = str(cpu)
        if sib not in seen:
            seen.add(sib)
            cores.append(cpu)
    return cores, len(allowed)


def cpu_quota():
    """CPUs' worth of time the cgroup of this process may use (cpu.max of cgroup v2, cfs_quota of v1; the smallest along the
    path to the root), or None when nothing limits it.  A GPU box of the pool shows 128 cores to sched_getaffinity and gives
    the container a share of them: threads beyond the quota are throttled, and spinning OpenMP waiters burn it."""
    best = None
    try:
        lines = open("/proc/self/cgroup").read().split()
    except OSError:
        lines = []
    paths = []
    for ln in lines:
        parts = ln.split(":", 2)
        if len(parts) != 3:
            continue
        if parts[0] == "0" and parts[1] == "":                      # v2
            paths.append(("/sys/fs/cgroup", parts[2], "cpu.max"))
        elif "cpu" in parts[1].split(","):                            # v1
            paths.append(("/sys/fs/cgroup/cpu", parts[2], None))
    for root, rel, v2file in paths or [("/sys/fs/cgroup", "/", "cpu.max")]:
        rel = rel.strip("/")
        while True:
            d = os.path.join(root, rel) if rel else root
            try:
                if v2file:
                    q, per = open(os.path.join(d, v2file)).read().split()[:2]
                    val = None if q == "max" else float(q) / float(per)
                else:
                    q = float(open(os.path.join(d, "cpu.cfs_quota_us")).read())
                    per = float(open(os.path.join(d, "cpu.cfs_period_us")).read())
                    val = None if q <= 0 else q / per
                if val is not None:
                    best = val if best is None else min(best, val)
            except (OSError, ValueError):
                pass
            if not rel:
                break
            rel = os.path.dirname(rel)
    return best


def cpu_baseline(sph, name, nsteps=20, windows=3):
    """The oracle (CPU restatement of the reference, reference flags -Ofast -march=native -fopenmp) timed on this
    host's cores on the same scene: the 'port' baseline (the reference itself cannot run > 65 534 particles: unsigned
    short indices, pi_sph_fluid.c:78-79).  A thread sweep, each leg in a fresh process so that OpenMP reads its settings:
    (a) 4 threads, the reference's shipped setting (:610); (b) as many threads as the container's CPU quota allows
    (cpu_quota; skipped when there is none or it equals another leg); (c) one thread per physical core this process may
    run on.  Legs up to the quota bind their threads (OMP_PROC_BIND=close OMP_PLACES=cores); legs beyond it wait passively
    (OMP_WAIT_POLICY=passive: a spinning waiter in the serial `omp single` parts :612-627 would burn the quota the working
    thread needs).  The particle arrays are first touched by the threads that work on them.  Median of `windows` windows of
    `nsteps` steps each; the best leg is the baseline, all are reported."""
    subprocess.check_call(["make", "-s", "-B", "-C", os.path.join(ROOT, "oracle"), "liborc_fast.so"])   # -march=native of THIS host
    cores, logical = physical_cores()
    quota = cpu_quota()
    eff = len(cores) if quota is None else max(1, min(len(cores), int(quota + 0.5)))      # cores' worth of time there really is
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    plan = [("4 threads (reference setting :610)", min(4, len(cores)))]
    if eff not in (plan[0][1], len(cores)):
        plan.append(("%d threads = the container's CPU quota" % eff, eff))
    if len(cores) != plan[0][1]:
        plan.append(("all %d physical cores" % len(cores), len(cores)))
    legs = []
    for label, threads in plan:
        over = threads > eff
        extra = {"OMP_WAIT_POLICY": "passive"} if over else {"OMP_PROC_BIND": "close", "OMP_PLACES": "cores"}
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), **extra)
        code = _CPU_LEG.format(root=ROOT, name=name, threads=threads, nsteps=nsteps, windows=windows)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, timeout=900)
        if r.returncode != 0:
            log("cpu_baseline leg failed:", r.stderr.decode()[-500:])
            continue
        d = json.loads(r.stdout.decode().strip().splitlines()[-1])
        rate = float(np.median(d["rates"]))
        n_fluid = d["n"]
        legs.append({"label": label + (", passive waiting (beyond the quota)" if over else ", bound to cores"), "threads": threads,
                     "timesteps_per_s": round(rate, 4), "value": round(rate * d["n"] / 1e6, 3), "windows": [round(x, 4) for x in d["rates"]]})
        log("cpu_baseline:", legs[-1])
    if not legs:          # (no leg ran: the GPU result is still worth a line)
        return None
    best = max(legs, key=lambda l: l["value"])
    res = {"value": best["value"], "unit": "Mparticle-steps/s", "timesteps_per_s": best["timesteps_per_s"],
            "cores": min(best["threads"], eff), "threads": best["threads"], "kind": "port", "legs": legs,
            "host": {"model": model, "physical_cores_visible": len(cores), "logical_cpus_visible": logical,
                     "cgroup_cpu_quota": None if quota is None else round(quota, 2), "effective_cores": eff},
            "sample": "median of %d windows of %d steps (after 2 warm-up) of the full %s scene, %d fluid particles; best of %d "
                      "legs (%s); host: %s, %d physical cores / %d logical CPUs visible, cgroup CPU quota %s -> %d effective cores; "
                      "`cores` = min(threads of the best leg, effective cores)"
                      % (windows, nsteps, name, n_fluid, len(legs), best["label"], model, len(cores), logical,
                         "none" if quota is None else "%.1f" % quota, eff)}
    try:          # the N > 1 runs of the same box quote it (they do not measure it again)
        with open(CPU_CACHE, "w") as fh:
            json.dump(dict(res, cached_from="the N = 1 run of bench.py on this host"), fh)
    except OSError:
        pass
    return res


CPU_CACHE = os.path.join(os.environ.get("TMPDIR", "/tmp"), "sph_bench_cpu_baseline_%d.json" % os.getuid())


def cached_cpu_baseline():
    try:
        with open(CPU_CACHE) as fh:
            return json.load(fh)
    except (OSError, ValueError):
        return None


def launch_ranks(args):
    """`python bench.py --gpus N` with no launcher around it: start one fresh process per GPU (this process has not
    touched the GPU: no HIP call, no torch import), each with its own RANK / LOCAL_RANK, rendezvous on 127.0.0.1.
    Rank 0's JSON line is forwarded; a failing rank fails the run.  (The torchrun form
    `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` works as well: the ranks then find
    WORLD_SIZE in their environment and never come here.)"""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    import tempfile
    procs = []
    out0 = tempfile.TemporaryFile()
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool (RCCL needs it)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))
    rcs = [None] * len(procs)
    while any(rc is None for rc in rcs):
        for k, pr in enumerate(procs):
            if rcs[k] is None:
                rcs[k] = pr.poll()
        if any(rc not in (None, 0) for rc in rcs):      # one rank failed: the others would wait for it forever
            for k, pr in enumerate(procs):
                if rcs[k] is None:
                    pr.kill()
                    rcs[k] = pr.wait()
        time.sleep(0.1)
    out0.seek(0)
    lines = [ln for ln in out0.read().decode(errors="replace").splitlines() if ln.startswith("{")]
    if any(rcs) or not lines:
        log("bench.py: ranks exited with", rcs)
        return 1
    os.write(1, (lines[-1] + "\n").encode())
    return 0


N1_CACHE = os.environ.get("SPH_BENCH_N1_CACHE") or os.path.join(os.environ.get("TMPDIR", "/tmp"), "sph_bench_n1_%d.json" % os.getuid())

# The one-GPU references of the N > 1 legs, by leg: the scene, the gravity and the EXACT window (warm-up steps, steps per window,
# windows: the rate is the median window) a leg is timed on.  A speed-up is only ever formed between two runs of the same window —
# cfg4 at rest (steps 50-650) runs at nearly twice the rate of its developed flow (steps 2000-2600).
STRONG_LEGS = {"at_rest": ("cfg4_at_rest", 50, 200, 3), "developed": ("cfg4_developed", 2000, 200, 3)}
if os.environ.get("SPH_BENCH_STRONG_DEVELOPED"):      # (rehearsals: "warmup,steps,windows" of the developed strong leg)
    STRONG_LEGS["developed"] = ("cfg4_developed",) + tuple(int(x) for x in os.environ["SPH_BENCH_STRONG_DEVELOPED"].split(","))


def n1_leg(tps, warmup, steps, windows):
    return {"timesteps_per_s": round(float(tps), 2), "warmup": int(warmup), "steps": int(steps), "windows": int(windows)}


def cached_n1():
    """what the N = 1 run of bench.py measured on this host, by leg (see write_n1_cache); {} when there is none"""
    try:
        with open(N1_CACHE) as fh:
            d = json.load(fh)
        return d.get("legs", {}) if d.get("schema") == 2 else {}
    except (OSError, ValueError, AttributeError):
        return {}


def write_n1_cache(legs):
    try:
        with open(N1_CACHE, "w") as fh:
            json.dump({"schema": 2, "written_by": "the N = 1 run of bench.py on this host", "legs": legs}, fh)
    except OSError:
        pass


def one_gpu_reference(sph, key, warmup, steps, windows, measure=True):
    """timesteps/s of ONE GPU (sph_step, the fastest one-GPU path) on the window a multi-GPU leg was timed on: the figure the N = 1
    run of this host cached for exactly that window, else measured here and now on device 0 (the multi-rank legs are over: the
    GPU is free) — never a figure from another window."""
    e = cached_n1().get(key)
    if e and (e.get("warmup"), e.get("steps"), e.get("windows")) == (warmup, steps, windows) and e.get("timesteps_per_s", 0) > 0:
        return {"timesteps_per_s": e["timesteps_per_s"], "reference": "cached by the N = 1 run of bench.py on this host (same window)",
                "window": [warmup, steps, windows]}
    if not measure:
        return None
    try:
        t0 = time.time()
        if key == "cfg2_window":
            r = run_single(sph, "cfg2", steps, warmup, windows=windows)
        else:
            r = run_single(sph, "cfg4", steps, warmup, profile_steps=5, tilt=True, windows=windows)
        log("one-GPU reference %s measured in %.1f s: %.2f steps/s" % (key, time.time() - t0, r["steps_per_s"]))
        return {"timesteps_per_s": round(r["steps_per_s"], 2), "reference": "measured in this run (sph_step on device 0, same window)",
                "window": [warmup, steps, windows], "window_timesteps_per_s": r["window_steps_per_s"]}
    except Exception as e_:      # (reported, never raised: the multi-GPU figures stand on their own)
        log("one-GPU reference %s failed: %r" % (key, e_))
        return {"timesteps_per_s": None, "reference": "unavailable: %r" % (e_,), "window": [warmup, steps, windows]}


def run_guarded(cmd, env, what):
    """A run of the C host under a time limit ($SPH_BENCH_LEG_TIMEOUT seconds, default 600: the longest leg — cfg4 developed, 2 600
    steps of 32 M particles over N ranks plus their creation — takes under a minute): a transport that never ran between GPUs may also
    never return, and a bench that hangs reports nothing at all.  The child is a session of its own (the host's launcher and its
    ranks): on a time-out the whole group is killed and the leg counts as failed (exit 124).  Returns (returncode, stdout bytes)."""
    import signal
    limit = float(os.environ.get("SPH_BENCH_LEG_TIMEOUT", "600"))
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, start_new_session=True)
    try:
        so, _ = p.communicate(timeout=limit)
        return p.returncode, so
    except subprocess.TimeoutExpired:
        log("bench.py: %s did not return within %.0f s: killed" % (what, limit))
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except OSError:
            pass
        so, _ = p.communicate()
        return 124, so


def c_host_cmd(host, scene, steps, warmup, transport, tilt, breakdown=30, windows=1):
    cmd = [host, "--scene", scene, "--steps", str(steps), "--warmup", str(warmup), "--windows", str(windows), "--transport", transport,
           "--breakdown", str(breakdown)]
    if tilt:
        cmd.append("--tilt")
    return cmd


def c_host_run(host, scene, steps, warmup, transport, tilt, world_env, tag, windows=1):
    """one run of the C multi-GPU host; returns (returncode, parsed JSON line or None, rank).  Under a launcher (world_env: this
    process is one rank of WORLD_SIZE) it runs the one rank; else it starts its own ranks."""
    cmd = c_host_cmd(host, scene, steps, warmup, transport, tilt, windows=windows)
    world, rank = world_env
    if world and transport == "rccl":          # torchrun started the ranks: this process is one of them
        # one file per job and leg: the launcher's run id (and port) name it; rank 0 removes it once every rank has joined
        job = "%s_%s_%d_%s" % (os.environ.get("TORCHELASTIC_RUN_ID", "none"), os.environ.get("MASTER_PORT", "0"), os.getppid(), tag)
        idfile = os.path.join(os.environ.get("TMPDIR", "/tmp"), "sph_bench_%d_%s.id" % (os.getuid(), "".join(ch for ch in job if ch.isalnum() or ch in "_-")))
        cmd += ["--ranks", str(world), "--rank", str(rank), "--id-file", idfile]
    else:
        cmd += ["--ranks", str(world or 1)]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rc, so = run_guarded(cmd, env, "slab_sph_fluid (%s, %s)" % (transport, tag))
    lines = [ln for ln in so.decode(errors="replace").splitlines() if ln.startswith("{")]
    return rc, (json.loads(lines[-1]) if lines else None)


def leg_summary(d, world, ref=None, weak=False):
    """the figures of one C-host run for the JSON line.  ref (one_gpu_reference): the one-GPU rate on the SAME window — speed-up
    against it (strong) or efficiency against it (weak: N x the particles at the one-GPU rate would be efficiency 1)"""
    out = {"value": round(d["mparticle_steps_per_s"], 2), "unit": "Mparticle-steps/s", "timesteps_per_s": round(d["ticks_per_s"], 2),
           "ms_per_step": d["ms_per_step"], "workload": d["workload"], "n_fluid": d["n_fluid"], "host": d["host"],
           "window": [d.get("warmup"), d.get("steps"), d.get("windows", 1)], "window_timesteps_per_s": d.get("window_ticks_per_s"),
           "particles_conserved": d["particles_conserved"], "neighbour_rebuilds": d["neighbour_rebuilds"],
           "halo_buffer_bytes": d.get("halo_buffer_bytes"),
           "kernel_ms": {"density_eos": d.get("rank0_density_ms"), "force_kick": d.get("rank0_force_ms")},
           # per rank: device, ranks its communicator counts, particles, and where a step's time goes (us per step over
           # `breakdown_steps` extra steps with an event at every phase boundary): a long begin / end = this rank's own kernels
           # (overloaded: re-balance), long reduce / exchange everywhere = waiting for the others / the interconnect
           "breakdown_steps": d.get("breakdown_steps"), "per_rank": d.get("per_rank")}
    if ref:
        out["reference"] = ref["reference"]
        out["reference_window"] = ref["window"]
        out["one_gpu_timesteps_per_s"] = ref["timesteps_per_s"]
        if ref["timesteps_per_s"] and list(ref["window"]) == [d.get("warmup"), d.get("steps"), d.get("windows", 1)]:
            if weak:
                out["weak_efficiency_vs_1gpu"] = round(d["ticks_per_s"] / ref["timesteps_per_s"], 4)
            else:
                out["speedup_vs_1gpu"] = round(d["ticks_per_s"] / ref["timesteps_per_s"], 3)
    return out


def transports_agree(a, b):
    """Did two runs of the C host — same scene, same window, different transports — compute the same flow?  Every particle
    owned once in both, and the reduced statistics of the final state (max rho, max speed: sph_stats over the ranks) equal to
    what the summation order of two runs can differ by.  (Ghost columns that arrive late, torn or not at all show up in the
    densities next to an interface first.)"""
    if not (a and b) or not (a.get("particles_conserved") and b.get("particles_conserved")):
        return False
    if any(a.get(k) != b.get(k) for k in ("n_fluid", "n_gpus", "steps", "warmup", "workload")):
        return False
    if not all(isinstance(x.get(k), (int, float)) for x in (a, b) for k in ("max_rho", "max_speed")):
        return False
    rho_ok = abs(a["max_rho"] - b["max_rho"]) <= 1e-3 * max(abs(a["max_rho"]), abs(b["max_rho"]), 1.0)
    speed_ok = abs(a["max_speed"] - b["max_speed"]) <= 1e-2 * max(a["max_speed"], b["max_speed"]) + 1e-3
    return bool(rho_ok and speed_ok)


def choose_headline(rccl_raw, peer_leg_result, margin=1.02):
    """--transport best: which run of the weak leg is the line's value.  The RCCL run unless the guarded peer run of the same
    window completed, agrees with it (transports_agree) and is faster by more than `margin`.  Returns ("rccl" | "peer", why)."""
    leg = peer_leg_result or {}
    raw = leg.get("raw")
    if leg.get("status") != "ok" or raw is None:
        return "rccl", "the peer run did not complete (%s)" % leg.get("status", "not run")
    if not transports_agree(rccl_raw, raw):
        return "rccl", "the peer run completed but does not agree with the RCCL run (particles / max rho / max speed)"
    if raw["ticks_per_s"] <= margin * rccl_raw["ticks_per_s"]:
        return "rccl", "the peer run agrees with the RCCL run and is not faster (%.1f against %.1f steps/s)" % (raw["ticks_per_s"], rccl_raw["ticks_per_s"])
    return "peer", ("the peer run of the same window agrees with the RCCL run (every particle owned once, max rho %.4g / %.4g, max speed "
                    "%.4g / %.4g) and is faster: %.1f against %.1f steps/s" % (raw["max_rho"], rccl_raw["max_rho"], raw["max_speed"],
                                                                                 rccl_raw["max_speed"], raw["ticks_per_s"], rccl_raw["ticks_per_s"]))


def run_c_host(sph, args):
    """N > 1 (default): the C multi-GPU host (pi-sph-fluid_amd/host/slab_sph_fluid.c: one process per GPU, halo
    exchange and rebuild-word reduction over RCCL, no torch in the loop).  Started here as N ranks, or — under torchrun
    — as the one rank this process stands for (the ncclUniqueId then travels through a file named after the job).
    The line's `value` is the WEAK-scaling run (2 000 000 particles per GPU: the cfg2 -> cfg3 family); beside it, under
    `scaling_detail`, the STRONG-scaling runs north_star asks for — cfg4 (32 000 000 particles under the scripted tilt, fixed) over
    the N ranks, once at rest (steps 50-650) and once developed (steps 2000-2600), each with its speed-up against the one-GPU rate
    on the SAME window (STRONG_LEGS; cached by this host's N = 1 run, else measured by rank 0 after the multi-rank legs) — and
    all of it once more over the peer transport (guarded: own process group, time limit).
    Default (round 6): the RCCL run is the line's value; the guarded peer run of the same windows is reported beside it
    (`peer_transport`) and never becomes the value.
    --transport best (opt-in): the RCCL run is the line's value unless the guarded peer run of the same window agrees with it and is
    faster (choose_headline): then that one is, and the line says so (`transport_used`, `value_is_max_of_two_runs`: a maximum of two
    runs is biased upwards).
    --transport host: the same step loop with POSIX shared memory between the ranks (they may share a device: a rehearsal of
    the N-rank code path on fewer GPUs, not a measurement of xGMI).  --transport auto: rccl, and if that run fails the peer
    run becomes the line's value (said so in `transport_used`); without it a failing transport fails the bench."""
    host = os.path.join(ROOT, "pi-sph-fluid_amd", "host", "slab_sph_fluid")
    if not os.path.exists(host):      # (normally built by __graft_entry__.build(); a fresh checkout builds it here)
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "pi-sph-fluid_amd"), "all"], stdout=sys.stderr)
    world = int(os.environ.get("WORLD_SIZE", "0"))
    rank = int(os.environ.get("RANK", "0")) if world else 0
    scene = {"cfg2": "dam", "dam": "dam", "cfg3": "cfg3", "cfg4": "cfg4"}[args.workload]
    best = args.transport == "best"      # (no fall-back: should the RCCL run fail, so does the bench — only --transport auto substitutes)
    auto = args.transport == "auto"
    transport = "rccl" if auto or best else args.transport
    under_launcher = bool(world)
    if under_launcher and transport != "rccl" and rank != 0:
        return          # (the shared-memory and peer transports start their own ranks: one launcher only)
    n_ranks = world or args.gpus

    def launch(scene_, steps, warmup, tr, tilt, tag, windows=1):
        if under_launcher and tr == "rccl":
            return c_host_run(host, scene_, steps, warmup, tr, tilt, (world, rank), tag, windows=windows)
        return _own_ranks(host, scene_, steps, warmup, tr, tilt, n_ranks, windows=windows)

    rc, d = launch(scene, args.steps, args.warmup, transport, args.workload == "cfg4", "weak")
    fallback = None
    if rc != 0 or (rank == 0 and d is None):
        log("bench.py: slab_sph_fluid (%s) exited with %d" % (transport, rc))
        if not (auto and n_ranks > 1):
            sys.exit(rc or 1)      # the requested transport failed: so does the bench (every rank: the launcher sees it)
        # --transport auto: rank 0 tries the same workload once over the peer transport, which needs no collective library, and
        # reports THAT, saying so; the other ranks of a launcher leave (their GPUs are free again)
        if rank != 0:
            return
        fallback = peer_leg(host, scene, n_ranks, args.steps, args.warmup, args.workload == "cfg4")
        if fallback.get("status") != "ok":
            log("bench.py: the peer transport did not complete either:", fallback)
            sys.exit(rc or 1)
        d, transport = fallback["raw"], "peer"
    # the strong-scaling legs (every rank of a launcher takes part; rank 0 reports)
    strong_raw = {}
    do_strong = scene == "dam" and not args.no_also and n_ranks > 1 and not fallback
    if do_strong:
        for name, (key, wu, st_, wn) in STRONG_LEGS.items():
            rc2, d2 = launch("cfg4", st_, wu, transport, True, "strong_" + name, windows=wn)
            strong_raw[name] = d2 if rc2 == 0 and d2 is not None else {"status": "failed (exit %d)" % rc2}
    if rank != 0:
        return
    peer_raw = {}
    if transport == "rccl" and n_ranks > 1 and not args.no_also:      # the guarded second leg of everything, before rank 0 takes the GPU for the references
        peer_raw["weak"] = peer_leg(host, scene, n_ranks, args.steps, args.warmup, args.workload == "cfg4")
        if scene == "dam":
            for name, (key, wu, st_, wn) in STRONG_LEGS.items():
                # (a transport that did not get through the small leg is not given two long ones to time out in: 3 minutes each)
                peer_raw[name] = (peer_leg(host, "cfg4", n_ranks, st_, wu, True, windows=wn) if peer_raw["weak"].get("status") == "ok"
                                  else {"status": "skipped: the weak leg over this transport did not complete (%s)" % peer_raw["weak"].get("status")})
    # --transport best: the faster of two runs that agree is the line's value (the other stays in the line)
    chosen, rccl_weak_raw = None, None
    if best and transport == "rccl" and n_ranks > 1 and "weak" in peer_raw:
        which, why = choose_headline(d, peer_raw["weak"])
        chosen = {"transport": which, "why": why}
        if which == "peer":
            rccl_weak_raw, d, transport = d, peer_raw["weak"]["raw"], "peer"
    # The one-GPU references, each on the window of its leg: from the N = 1 run's cache when the windows agree, else measured now
    # (the multi-rank legs are over, device 0 is free).  A one-slab run (the slab path's own overhead) has nothing to scale against.
    refs = {}
    if n_ranks > 1:
        if scene == "dam":
            refs["weak"] = one_gpu_reference(sph, "cfg2_window", args.warmup, args.steps, 1)
        if do_strong or peer_raw:
            for name, (key, wu, st_, wn) in STRONG_LEGS.items():
                if "ticks_per_s" in (strong_raw.get(name) or {}) or (peer_raw.get(name) or {}).get("status") == "ok":
                    refs[name] = one_gpu_reference(sph, key, wu, st_, wn)
    strong = None
    if do_strong:
        strong = {name: (leg_summary(r, n_ranks, refs.get(name)) if "ticks_per_s" in r else r) for name, r in strong_raw.items()}
    n_total, tps = d["n_fluid"], d["ticks_per_s"]
    step_gbs = sph.STEP_ALGO_BYTES * n_total * tps / 1e9 / n_ranks
    # the dominant kernel (the force pass, which also integrates) of rank 0's slab against the HBM roofline of ITS GPU:
    # 80 algorithmic bytes per particle the launch covers (owned + ghosts), live duration from the C host
    force_ms, n_loc = d.get("rank0_force_ms", 0.0), d.get("rank0_local", 0)
    force_gbs = sph.KERNEL_ALGO_BYTES["force_kick"] * n_loc / (force_ms * 1e-3) / 1e9 if force_ms > 0 else None
    how = {"rccl": ("ncclSend/ncclRecv pair", "4-byte ncclAllReduce(max)"),
           "host": ("shared-memory mailbox (REHEARSAL transport: the ranks may share a GPU)", "host max-reduction"),
           "peer": ("store into the neighbours' hipIpc-mapped memory (no collective library)", "exchange of flag words")}[transport]
    weak = leg_summary(d, n_ranks, refs.get("weak"), weak=True)
    out = {
        "metric": "SPH Mparticle-steps/sec (N_fluid x timesteps/sec / 1e6)",
        "value": round(d["mparticle_steps_per_s"], 2), "unit": "Mparticle-steps/s", "timesteps_per_s": round(tps, 2),
        "n_gpus": n_ranks, "steps": args.steps, "warmup": args.warmup, "ms_per_step": d["ms_per_step"],
        "higher_is_better": True, "scaling": "weak" if scene == "dam" else "strong", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "%s: %d fluid + %d boundary particles" % (d["workload"], n_total, d["n_boundary"]),
                   "n_fluid": n_total, "n_boundary": d["n_boundary"],
                   "parallelism": "%d x-slabs, one process per GPU, C host (slab_sph_fluid): 2-column halo + migration in one "
                                  "%s per neighbour per step, %s of the rebuild word" % (n_ranks, how[0], how[1])},
        "transport": transport,
        "particles_conserved": d["particles_conserved"],
        "neighbour_rebuilds_per_step": round(d["neighbour_rebuilds"] / max(args.steps + args.warmup, 1), 4),
        "kernel_ms": {"density_eos": d.get("rank0_density_ms"), "force_kick": d.get("rank0_force_ms")},
        "roofline": {"bound": "hbm", "kernel": "force_kick (rank 0's slab: %d particles incl. ghosts)" % n_loc,
                     "achieved": round(force_gbs, 1) if force_gbs else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(force_gbs / HBM_PEAK_GBS, 4) if force_gbs else None, "traffic": None,
                     "step_achieved": round(step_gbs, 1), "step_frac": round(step_gbs / HBM_PEAK_GBS, 4),
                     "step_unit": "GB/s per GPU (whole step, 152 B per particle-step)"},
        "per_rank": d.get("per_rank"), "breakdown_steps": d.get("breakdown_steps"), "halo_buffer_bytes": d.get("halo_buffer_bytes"),
        # weak: the line's own run (2 000 000 particles per GPU); strong: cfg4 (32 000 000 particles, fixed) over the same ranks,
        # at rest and developed, each against the one-GPU rate on its own window
        "scaling_detail": {"weak": weak, "strong": strong},
        "cpu_baseline": cached_cpu_baseline(),      # measured by the N = 1 run on this host (None if there was none)
        "box": box_calibration(sph),                # what device 0 of this node delivers (after the multi-rank legs: the GPU is free)
    }
    if fallback:
        out["transport_used"] = "peer (--transport auto: the RCCL run of this bench did not complete: exit %d)" % rc
    if chosen:
        out["transport_choice"] = chosen      # --transport best: which of the two weak-leg runs is the line's value, and why
        out["value_is_max_of_two_runs"] = True      # (a maximum of two runs is biased upwards: opt-in only, and said)
        out["scaling_detail"]["strong_transport"] = "rccl"      # (the strong legs under scaling_detail are the RCCL runs; the peer ones: peer_transport.strong)
        if rccl_weak_raw is not None:
            out["transport_used"] = "peer (--transport best: %s)" % chosen["why"]
            out["rccl_transport"] = {"weak": leg_summary(rccl_weak_raw, n_ranks, refs.get("weak"), weak=True)}
    if peer_raw:
        leg = peer_raw["weak"]
        raw = leg.pop("raw", None)
        out["peer_transport"] = leg
        if raw is not None:
            out["peer_transport"]["weak"] = leg_summary(raw, n_ranks, refs.get("weak"), weak=True)
        if scene == "dam":
            out["peer_transport"]["strong"] = {}
            for name in STRONG_LEGS:
                leg2 = peer_raw[name]
                raw2 = leg2.pop("raw", None)
                out["peer_transport"]["strong"][name] = leg_summary(raw2, n_ranks, refs.get(name)) if raw2 is not None else leg2
    emit(out)


def _own_ranks(host, scene, steps, warmup, transport, tilt, n_ranks, windows=1):
    """the C host started as its own launcher (it forks its ranks before anything touches a GPU)"""
    cmd = c_host_cmd(host, scene, steps, warmup, transport, tilt, windows=windows) + ["--ranks", str(n_ranks)]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    rc, so = run_guarded(cmd, env, "slab_sph_fluid --ranks %d (%s)" % (n_ranks, transport))
    lines = [ln for ln in so.decode(errors="replace").splitlines() if ln.startswith("{")]
    return rc, (json.loads(lines[-1]) if lines else None)


def peer_leg(host, scene, world, steps, warmup, tilt, windows=1):
    """After the RCCL run (whose numbers are the line's `value`): the same workload once more with --transport peer — the
    step's traffic as stores into hipIpc-mapped peer memory and flag words, three small kernels instead of RCCL's
    all-reduce and send / receive — started by the C host's own launcher, in its own process group, under a time limit.
    A one-GPU pool cannot exercise that transport between GPUs; this leg is how it gets its first run over xGMI without
    putting the headline at risk.  Whatever happens here is reported, never raised."""
    import signal
    cmd = c_host_cmd(host, scene, steps, warmup, "peer", tilt, windows=windows) + ["--ranks", str(world)]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    try:
        p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, start_new_session=True)
    except OSError as e:
        return {"status": "not started: %s" % e}
    try:
        so, se = p.communicate(timeout=float(os.environ.get("SPH_BENCH_PEER_TIMEOUT", "180")))
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)      # (the launcher and the ranks it started: this leg's own process group)
        except OSError:
            pass
        p.communicate()
        return {"status": "timed out"}
    lines = [ln for ln in so.decode(errors="replace").splitlines() if ln.startswith("{")]
    if p.returncode != 0 or not lines:
        return {"status": "failed (exit %d)" % p.returncode, "stderr_tail": se.decode(errors="replace")[-600:]}
    d = json.loads(lines[-1])
    return {"status": "ok", "raw": d}


def slab_overhead(sph, out):
    """What the SLAB path costs before any neighbour exists, at the two slab sizes of the multi-GPU runs — 2 000 000 particles
    (the weak family's slab = cfg2) and 4 000 000 (one eighth of cfg4 as a tank of its own, under the tilt trace) — one rank of
    the C host on this GPU against sph_step on the same particles, same windows: the lean step (sph_slab_step: head | update or
    rebuild | density | force, four kernels: what a rank runs over the peer transport), the three-call step (six kernels), and
    the three-call step with the per-step calls of either transport issued to the rank itself (--selfcomm: RCCL's all-reduce +
    grouped send / receive; the peer transport's three kernels against its own block).  From the 4M figures: the most an
    8-GPU strong-scaling run of cfg4 can gain — t(32M on one GPU) / t(its slab through the slab path), with perfect overlap of
    everything between the ranks."""
    host = os.path.join(ROOT, "pi-sph-fluid_amd", "host", "slab_sph_fluid")

    def c_host(scene, warmup, steps, windows, tilt, extra):
        cmd = [host, "--ranks", "1", "--scene", scene, "--steps", str(steps), "--warmup", str(warmup), "--windows", str(windows)] + (["--tilt"] if tilt else []) + extra
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        lines = [ln for ln in r.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            raise RuntimeError("slab_sph_fluid %s: exit %d: %s" % (" ".join(extra), r.returncode, r.stderr.decode(errors="replace")[-300:]))
        d = json.loads(lines[-1])
        return {"timesteps_per_s": round(d["ticks_per_s"], 2), "window_timesteps_per_s": d.get("window_ticks_per_s"), "host": d["host"],
                "rebuilds_per_step_whole_run": round(d["neighbour_rebuilds"] / max(warmup + windows * steps, 1), 4)}

    variants = [("lean", "sph_slab_steps as the C host runs them by default (--lean-spec 2): the FUSED speculative lean step, 3 kernels per step — density with the slab head's "
                         "work and the criterion's jobs in its launch | gate | force —, graphs of up to 16 steps, no neighbour", []),
                ("lean_spec4", "the speculative lean step as four launches (--lean-spec 1: head | density with the criterion's jobs | gate | force)", ["--lean-spec", "1"]),
                ("lean_plain", "the plain lean step (--lean-spec 0: the criterion's boxes in the head kernel, no verification), graphs of up to 16 steps", ["--lean-spec", "0"]),
                ("three_call", "the three-call step: 6 kernels, no neighbour", ["--lean", "0"]),
                ("three_call_rccl_selfcomm", "three-call + the step's RCCL calls (all-reduce, grouped send / receive) to the rank itself",
                 ["--lean", "0", "--selfcomm", "--transport", "rccl"]),
                ("three_call_peer_selfcomm", "three-call + the peer transport's three kernels against the rank's own block",
                 ["--lean", "0", "--selfcomm", "--transport", "peer"])]
    res = {"workload": "slab_overhead: one slab through the C host vs sph_step on the same particles", "sizes": {}}
    # 2M: cfg2 in the 8(d) protocol — sph_step's figure is the `sustained` entry of this line
    one = out.get("sustained", {}).get("timesteps_per_s")
    size = {"n_fluid": 2000000, "window": [200, 1000, 5], "sph_step_timesteps_per_s": one}
    for key, what, extra in variants:
        r = c_host("dam", 200, 1000, 5, False, extra)
        size[key] = dict(r, what=what, vs_sph_step=round(r["timesteps_per_s"] / one, 4) if one else None)
        log("slab_overhead 2M %s: %s" % (key, size[key]))
    res["sizes"]["2M (the weak family's slab = cfg2)"] = size
    # 4M: one of cfg4's eight slabs as a tank of its own, tilt trace, cfg4's at-rest window
    key4, wu, st_, wn = STRONG_LEGS["at_rest"]
    r1 = run_single(sph, "cfg4_slab", st_, wu, profile_steps=5, tilt=True, windows=wn)
    size = {"n_fluid": r1["n_fluid"], "window": [wu, st_, wn], "sph_step_timesteps_per_s": round(r1["steps_per_s"], 2),
            "sph_step_window_timesteps_per_s": r1["window_steps_per_s"]}
    for key, what, extra in variants:
        r = c_host("cfg4slab", wu, st_, wn, True, extra)
        size[key] = dict(r, what=what, vs_sph_step=round(r["timesteps_per_s"] / r1["steps_per_s"], 4))
        log("slab_overhead 4M %s: %s" % (key, size[key]))
    t32 = next((e["timesteps_per_s"] for e in out.get("also", []) if e.get("cache_key") == key4), None)
    if t32:
        size["cfg4_one_gpu_timesteps_per_s"] = t32
        size["strong_scaling_upper_bound_8_gpus"] = {k: round(size[k]["timesteps_per_s"] / t32, 2) for k, _w, _e in variants}
        size["strong_scaling_upper_bound_8_gpus"]["sph_step_on_the_slab"] = round(r1["steps_per_s"] / t32, 2)
    res["sizes"]["4M (one of cfg4's eight slabs, tilt trace, at rest)"] = size
    # 4M DEVELOPED (round 6): the same tank once its lattice has fallen and bounced (the window of cfg4's developed leg) — the regime that
    # decides the strong leg: rebuilds cost milliseconds there and a slab context rebuilds more often than sph_step (no verification by
    # default, the absolute criterion next to ghosts).  One slab through the C host (lean, graphed) against sph_step on the same particles,
    # and the bound t(32 M developed on one GPU) / t(its slab): what 8 GPUs can reach at most with everything between them hidden.
    keyd, wu, st_, wn = STRONG_LEGS["developed"]
    try:
        r1 = run_single(sph, "cfg4_slab", st_, wu, profile_steps=5, tilt=True, windows=wn)
        total = wu + wn * st_
        size = {"n_fluid": r1["n_fluid"], "window": [wu, st_, wn], "sph_step_timesteps_per_s": round(r1["steps_per_s"], 2),
                "sph_step_window_timesteps_per_s": r1["window_steps_per_s"], "sph_step_rebuilds_per_step_whole_run": round(r1["rebuilds"] / max(total + 5, 1), 4),
                "sph_step_window_rebuilds_per_step": r1["window_rebuilds_per_step"]}
        for key, what, extra in variants[:4]:
            d = c_host("cfg4slab", wu, st_, wn, True, extra)
            size[key] = dict(d, what=what, vs_sph_step=round(d["timesteps_per_s"] / r1["steps_per_s"], 4))
            log("slab_overhead 4M developed %s: %s" % (key, size[key]))
        t32 = next((e["timesteps_per_s"] for e in out.get("also", []) if e.get("cache_key") == keyd), None)
        if t32:
            size["cfg4_one_gpu_timesteps_per_s"] = t32
            size["strong_scaling_upper_bound_8_gpus"] = {k: round(size[k]["timesteps_per_s"] / t32, 2) for k, _w, _e in variants[:4]}
            size["strong_scaling_upper_bound_8_gpus"]["sph_step_on_the_slab"] = round(r1["steps_per_s"] / t32, 2)
        res["sizes"]["4M developed (the same tank, steps %d-%d: the lattice has fallen)" % (wu, total)] = size
    except Exception as e_:      # (reported, never raised)
        log("slab_overhead 4M developed failed: %r" % (e_,))
        res["sizes"]["4M developed"] = {"status": "failed: %r" % (e_,)}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-also", action="store_true", help="skip the secondary measurements")
    ap.add_argument("--save-state", default=None, help="N = 1: write the state after the warm-up (particles + accelerations, .npz)")
    ap.add_argument("--load-state", default=None, help="N = 1: start from a state written by --save-state (then --warmup, then the timed steps)")
    ap.add_argument("--tilt", action="store_true", help="N = 1: gravity from the scripted tilt trace (cfg4 is defined with it)")
    ap.add_argument("--transport", default=os.environ.get("SPH_SLAB_TRANSPORT", "rccl"), choices=["best", "rccl", "host", "peer", "auto"],
                    help="N > 1: halo transport. rccl = RCCL over xGMI, one GPU per rank; host = host-staged (C host: POSIX "
                         "shared memory; python host: gloo): a rehearsal, all ranks may share one device; peer = stores into "
                         "hipIpc-mapped peer memory + flag words (C host), no collective library on the step path; auto = rccl, and "
                         "should that run fail, peer (reported as such) — without it a failing transport fails the bench; rccl is the default "
                         "(round 6: the line's value is the RCCL run, the guarded peer run is reported beside it under peer_transport); best (opt-in) = rccl, and "
                         "where the guarded peer run of the same window completes, agrees with the RCCL run (particles, max rho, max "
                         "speed) and is faster, ITS figure is the line's value (both are reported; `transport_used` and `value_is_max_of_two_runs` say so)")
    ap.add_argument("--lib", default=None, help="A/B measurements: load this build of libsph_hip.so instead of the in-tree one")
    ap.add_argument("--skin", type=float, default=None, help="Verlet skin as a fraction of 2H (default: the library's)")
    ap.add_argument("--slab-host", default="c", choices=["c", "python"],
                    help="N > 1: c = the C host over RCCL (slab_sph_fluid), python = torch.distributed (bench_slab.py)")
    ap.add_argument("--rebalance", action="store_true",
                    help="N > 1, python host: re-balance the slab column ranges once after the warm-up (outside the timed region)")
    ap.add_argument("--slabs-on-one-gpu", action="store_true",
                    help="run the N = 1 workload through the slab path (one slab, no exchange): its overhead")
    args = ap.parse_args()

    sph = importlib.import_module("pi-sph-fluid_amd")
    if not (os.path.exists(sph.LIB_HIP) and os.path.exists(sph.LIB_HOST)):
        sph.build()      # before any rank starts
    if args.lib:
        sph.LIB_HIP = os.path.abspath(args.lib)

    slabbed = args.gpus > 1 or int(os.environ.get("WORLD_SIZE", "1")) > 1 or args.slabs_on_one_gpu
    if slabbed and args.slab_host == "c":
        quiet_stdout()
        run_c_host(sph, args)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    quiet_stdout()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 or world > 1 or args.slabs_on_one_gpu:
        from bench_slab import run_slabs          # one process per GPU over RCCL (torch.distributed)
        run_slabs(sph, args, emit)
        return

    box = box_calibration(sph)      # before the timed region (it also brings a fresh box to its running clocks)
    log("box:", json.dumps(box))
    res = run_single(sph, args.workload, args.steps, args.warmup, skin=args.skin, tilt=args.tilt, load_state=args.load_state,
                     save_state=args.save_state)
    log("primary:", json.dumps(res))
    out = {
        "metric": "SPH Mparticle-steps/sec (N_fluid x timesteps/sec / 1e6)",
        "value": round(res["mparticle_steps_per_s"], 2),
        "unit": "Mparticle-steps/s",
        "timesteps_per_s": round(res["steps_per_s"], 2),
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(res["ms_per_step"], 5),
        # what ran between the warm-up steps and the timed window: back-to-back launches of the two walkers on the live state (their
        # duration at the start of the window: kernel_ms.*_at_begin_end); they also bring a fresh box to its running clocks
        "pre_timing_launches": 2 * PRE_REPS if args.warmup > 0 and not args.load_state else 0,
        # (round 3 chose a cheaper set of step graphs at host synchronisation points while the fluid was at rest; gone: the step
        # is the same three launches in every regime — density with the rebuild criterion inside, gate / rebuild, force)
        "rest_mode": False, "launches_per_step": 3,
        "higher_is_better": True, "scaling": None, "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s: %d fluid + %d boundary particles, %s" %
                   (res["workload"], res["n_fluid"], res["n_boundary"],
                    "dam break, box 1200 x 60 m" if res["workload"] == "cfg2" else "see SURVEY.md 8d"),
                   "n_fluid": res["n_fluid"], "n_boundary": res["n_boundary"], "grid_cells": res["grid_cells"],
                   "parallelism": "1 GPU"},
        "kernel_ms": {k: (v if isinstance(v, list) else round(v, 5)) for k, v in res["kernel_ms"].items()},
        "neighbour_rebuilds_per_step": round(res["timed_rebuilds_per_step"], 4), "skin_fraction_of_2h": round(res["skin_frac"], 4),
        "skin_min_fraction_of_2h": round(res["skin_min_frac"], 4), "skin_at_end_fraction_of_2h": round(res["skin_now"], 4),
        # pairs of box groups checked particle by particle per step instead of rebuilding, and who asked for the rebuilds of the
        # timed region: [box pairs that could not be verified, verification found a missing pair, drift cap, rest mode]
        "verified_group_pairs_per_step": round(res["verified_pairs_per_step"], 1), "rebuild_requests": res["rebuild_requests"],
        # pairs appended to the lists in the timed region instead of a rebuild, and repairs that were not possible: [done, partner not
        # staged within reach, no free byte in the lane's rows, queue of repaired tiles full]
        "list_repairs": res["list_repairs"],
        # (the PMC traffic of the regime the window is in: steps 4000+ = the developed flow; the first ~100 steps = the fluid at
        # rest with the smallest skin — the driver's `--steps 20 --warmup 5`; otherwise the early collapse of steps 100-400)
        "roofline": vs_copy(roofline(sph, res, None if args.workload != "cfg2" else "cfg2_developed" if args.warmup >= 3000 else
                                     "cfg2_at_rest" if args.warmup + args.steps <= 100 else None), box),
        "box": box,
    }

    def also_entry(label, r, traffic_key, cache_key=None):
        """one secondary measurement in SURVEY.md 8(d)'s protocol: warm-up, then the MEDIAN of several windows — whatever
        --steps says (the headline above is exactly --steps steps after --warmup)"""
        rf = roofline(sph, r, traffic_key)
        return {"workload": label, "protocol": "8d", "cache_key": cache_key, "window": [r["warmup"], r["window_steps"], r["windows"]],
                "protocol_detail": "%d warm-up steps, median of %d windows of %d steps" % (r["warmup"], r["windows"], r["window_steps"]),
                "value": round(r["mparticle_steps_per_s"], 2), "unit": "Mparticle-steps/s",
                "timesteps_per_s": round(r["steps_per_s"], 2), "ms_per_step": round(r["ms_per_step"], 5),
                "window_timesteps_per_s": r["window_steps_per_s"], "window_rebuilds_per_step": r["window_rebuilds_per_step"],
                "kernel_ms": {k: (v if isinstance(v, list) else round(v, 5)) for k, v in r["kernel_ms"].items()},
                "neighbour_rebuilds_per_step": round(r["timed_rebuilds_per_step"], 4),
                "verified_group_pairs_per_step": round(r["verified_pairs_per_step"], 1), "rebuild_requests": r["rebuild_requests"],
                "list_repairs": r["list_repairs"], "max_speed": round(r["max_speed"], 2), "device_mb": round(r["device_mb"], 1), "direct_tiles": r["direct_tiles"],
                "step_frac": rf["step_frac"], "step_frac_executed": rf["step_frac_executed"], "roofline": rf}

    if not args.no_also and args.workload == "cfg2" and args.skin is None and not args.load_state:
        # The same window with the skin FIXED at round 2's 0.15 x 2H: the like-for-like figure for the two list walkers
        # (the adaptive skin of the headline is near 0.30 at the end of the window: longer lists, fewer rebuilds, a faster
        # step but slower kernels — their times in `roofline` / `kernel_ms` above are at THAT skin).
        r = run_single(sph, "cfg2", args.steps, args.warmup, skin=0.15)
        log("fixed skin:", json.dumps(r))
        rf = roofline(sph, r, "cfg2_fixed_skin_0.15")
        out["kernels_at_fixed_skin_0.15"] = {
            "what": "the headline window (same steps, same warm-up) with sph_params.skin = skin_min = 0.15, round 2's setting",
            "timesteps_per_s": round(r["steps_per_s"], 2), "neighbour_rebuilds_per_step": round(r["timed_rebuilds_per_step"], 4),
            "density_eos_ms": round(r["kernel_ms"]["density_eos"], 5), "force_kick_ms": round(r["kernel_ms"]["force_kick"], 5),
            "roofline": {k: rf[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algo_bytes_per_launch", "kernel_ms")}}
    if not args.no_also and args.workload == "cfg2":
        out["also"] = []
        # the headline scene itself, in the protocol: 200 warm-up steps, 5 windows of 1000
        r = run_single(sph, "cfg2", 1000, 200, skin=args.skin, windows=5)
        log("also:", json.dumps(r))
        out["also"].append(also_entry("cfg2: %d fluid + %d boundary, dam break, box 1200 x 60 m" % (r["n_fluid"], r["n_boundary"]), r, "cfg2", "cfg2_8d"))
        # the figure to anchor on: the same scene in SURVEY 8(d)'s protocol (the headline above is whatever window --steps /
        # --warmup ask for: the driver's 20 steps after 5 are a fluid still at rest)
        e = out["also"][-1]
        out["sustained"] = {"what": "cfg2 in SURVEY 8(d)'s protocol: 200 warm-up steps, median of five windows of 1000 steps",
                            "value": e["value"], "unit": e["unit"], "timesteps_per_s": e["timesteps_per_s"],
                            "window_timesteps_per_s": e["window_timesteps_per_s"], "window_rebuilds_per_step": e["window_rebuilds_per_step"],
                            "step_frac": e["step_frac"], "step_frac_executed": e["step_frac_executed"]}
        if box.get("copy_gbs"):      # ... and against what a streaming copy reaches on THIS box
            out["sustained"]["step_frac_vs_copy"] = round(e["step_frac"] * HBM_PEAK_GBS / box["copy_gbs"], 4)
            out["sustained"]["timesteps_per_s_per_copy_tbs"] = round(e["timesteps_per_s"] / (box["copy_gbs"] / 1e3), 1)
        r = run_single(sph, "cfg1", 1000, 200, skin=args.skin, windows=5)
        log("also:", json.dumps(r))
        out["also"].append(also_entry("cfg1: %d fluid + %d boundary, drop on dry surface, box 409.6 x 204.8 m" % (r["n_fluid"], r["n_boundary"]), r, "cfg1"))
        # the same dam break once the flow is developed (splashes, |v| ~ 25 m/s): steps 4000 - 9000; the neighbour
        # structure is rebuilt three times as often as in the first thousand steps
        r = run_single(sph, "cfg2", 1000, 4000, skin=args.skin, windows=5)
        log("also:", json.dumps(r))
        out["also"].append(also_entry("cfg2, developed flow: steps 4000-9000 of the same run", r, "cfg2_developed"))
        # cfg4 as BASELINE.json states it, on ONE GPU: 32 000 000 particles (6.8 GB of device memory: the HBM-resident
        # point, nothing fits the 256 MiB Infinity Cache) under the scripted tilt trace; 3 windows of 200 steps
        r = run_single(sph, "cfg4", STRONG_LEGS["at_rest"][2], STRONG_LEGS["at_rest"][1], profile_steps=5, skin=args.skin, tilt=True, windows=STRONG_LEGS["at_rest"][3])
        log("also:", json.dumps(r))
        out["also"].append(also_entry("cfg4 on one GPU: %d fluid + %d boundary, box 2400.6 x 150 m, scripted tilt gravity"
                                      % (r["n_fluid"], r["n_boundary"]), r, "cfg4", "cfg4_at_rest" if args.skin is None else None))
        # ... and the same tank once the un-compressed lattice has fallen onto the floor (75 m of water: |v| to 60 m/s at the
        # bounce): steps 2000 - 2600.  The HBM-resident DEVELOPED point: nothing fits the Infinity Cache and the lists are rebuilt
        r = run_single(sph, "cfg4", 200, 2000, profile_steps=5, skin=args.skin, tilt=True, windows=3)      # (= STRONG_LEGS["developed"] by default)
        log("also:", json.dumps(r))
        out["also"].append(also_entry("cfg4 on one GPU, developed flow: steps 2000-2600 of the same run", r, "cfg4_developed",
                                      "cfg4_developed" if args.skin is None else None))
    if not args.no_also and args.workload == "cfg2" and args.skin is None and not args.load_state and not args.lib:
        try:
            out["also"].append(slab_overhead(sph, out))
        except Exception as e_:      # (reported, never raised)
            log("slab_overhead failed: %r" % (e_,))
            out["also"].append({"workload": "slab_overhead", "status": "failed: %r" % (e_,)})
    if out.get("also") and args.skin is None and not args.load_state and not args.lib:
        # the N > 1 runs of this host quote these (weak efficiency, strong speed-ups): by leg, each with its window; only what the
        # library's defaults measured (a fixed skin, a checkpoint or another build are A/B runs)
        legs = {"cfg2_window": n1_leg(out["timesteps_per_s"], args.warmup, args.steps, 1)}
        for e in out["also"]:
            if e.get("cache_key"):
                legs[e["cache_key"]] = n1_leg(e["timesteps_per_s"], *e["window"])
        write_n1_cache(legs)
    if not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(sph, args.workload)
        if out["cpu_baseline"]:      # the sustained figure when there is one (SURVEY 8d's protocol), not whatever window --steps names
            gpu = out["sustained"]["value"] if out.get("sustained") else out["value"]
            out["gpu_over_cpu"] = round(gpu / out["cpu_baseline"]["value"], 1)
            out["gpu_over_cpu_basis"] = "sustained" if out.get("sustained") else "the headline window"
    emit(out)


if __name__ == "__main__":
    main()

"""bench.py --gpus N starts its own ranks (no torchrun needed) and never hangs: one process per rank, rendezvous on
127.0.0.1, rank 0's JSON line forwarded, a failing rank fails the whole run.

CPU (no GPU in the container): the ranks rendezvous over gloo, then slab creation fails LOUDLY (the product has no CPU
path) and the launcher returns non-zero within seconds.  GPU box: two ranks share the one MI355X through the host-staged
transport and the run conserves every particle."""
import json
import os
import subprocess
import sys
import time

import pytest

from conftest import ROOT

CMD = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--transport", "host", "--slab-host", "python",
       "--steps", "6", "--warmup", "3"]


def test_self_launch_fails_loudly_without_gpu(sph):
    if sph.hip_lib().sph_device_count() > 0:
        pytest.skip("a GPU is present: covered by test_self_launch_two_ranks_on_one_gpu")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    t0 = time.time()
    r = subprocess.run(CMD, capture_output=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode != 0
    assert b"no HIP device" in r.stderr and b"no CPU path" in r.stderr, r.stderr[-2000:]
    assert not [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]      # no result line
    assert time.time() - t0 < 300                                                       # and no rendezvous hang


@pytest.mark.gpu
def test_self_launch_two_ranks_on_one_gpu(sph):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run(CMD, capture_output=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 6 and out["particles_conserved"] is True
    assert out["config"]["n_fluid"] == 4000000 and out["value"] > 0 and out["scaling"] == "weak"
    # and with a re-balancing of the column ranges after the warm-up (collectives over gloo, slab contexts re-created)
    r = subprocess.run(CMD + ["--rebalance"], capture_output=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][0])
    assert out["particles_conserved"] is True and b"re-balanced" in r.stderr


def test_c_host_path_fails_loudly_without_gpu(sph):
    """the default N > 1 path (the C host over RCCL) through bench.py: two ranks, no GPU -> non-zero, at once"""
    if sph.hip_lib().sph_device_count() > 0:
        pytest.skip("a GPU is present")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2"],
                       capture_output=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0 and b"no HIP device available" in r.stderr
    assert not [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]


@pytest.mark.gpu
def test_c_host_path_one_slab(sph):
    """bench.py's N > 1 leg with one rank: the C host over RCCL (ncclCommInitRank with one rank), JSON line of the contract"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--slabs-on-one-gpu", "--steps", "40",
                        "--warmup", "10"], capture_output=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["particles_conserved"] is True and out["config"]["n_fluid"] == 2000000
    assert out["unit"] == "Mparticle-steps/s" and out["value"] > 0 and "roofline" in out
    # and under a torchrun-style environment (the driver's launch form): the rank comes from RANK / WORLD_SIZE
    env2 = dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29777")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--slabs-on-one-gpu", "--steps", "20",
                        "--warmup", "5"], capture_output=True, timeout=600, env=env2, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["particles_conserved"] is True


@pytest.mark.gpu
def test_c_host_two_ranks_share_the_gpu_through_bench(sph):
    """bench.py --gpus 2 --transport host: the C host's N > 1 step loop (two processes, POSIX shared memory between them) on
    the one GPU of this box; the line carries the dominant kernel's roofline and the kernels' live durations"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--transport", "host", "--steps", "30",
                        "--warmup", "10"], capture_output=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["particles_conserved"] is True and out["config"]["n_fluid"] == 4000000
    assert out["scaling"] == "weak" and "shared-memory" in out["config"]["parallelism"]
    assert out["roofline"]["kernel"].startswith("force_kick") and 0 < out["roofline"]["frac"] < 1
    assert out["kernel_ms"]["force_kick"] > 0 and out["kernel_ms"]["density_eos"] > 0
    assert "cpu_baseline" in out


@pytest.mark.gpu
def test_c_host_two_ranks_over_peer_mapped_memory_through_bench(sph):
    """bench.py --gpus 2 --transport peer: two processes on the one GPU of this box that store their halo buffers into each
    other's hipIpc-mapped memory; and bench.peer_leg(), the guarded second run that an N > 1 RCCL bench adds to its line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--transport", "peer", "--steps", "30",
                        "--warmup", "10"], capture_output=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 2 and out["particles_conserved"] is True and out["config"]["n_fluid"] == 4000000
    assert "hipIpc-mapped" in out["config"]["parallelism"] and "peer_transport" not in out
    sys.path.insert(0, ROOT)
    import bench
    leg = bench.peer_leg(os.path.join(ROOT, "pi-sph-fluid_amd", "host", "slab_sph_fluid"), "dam", 2, 30, 10, False)
    assert leg["status"] == "ok" and leg["raw"]["particles_conserved"] is True and leg["raw"]["ticks_per_s"] > 0 and "peer-mapped" in leg["raw"]["host"], leg
    # the default transport with more ranks than GPUs: the RCCL run refuses (one rank per GPU) and the bench FAILS — a transport
    # that was asked for and did not run is not replaced silently ...
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "30", "--warmup", "10"],
                       capture_output=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode != 0 and b"RCCL does not share a device" in r.stderr
    assert not [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    # ... unless the caller opts in: --transport auto falls back to the peer transport and says so
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--transport", "auto", "--steps", "30", "--warmup", "10"],
                       capture_output=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 2 and out["particles_conserved"] is True and out["transport_used"].startswith("peer (--transport auto")
    assert out["transport"] == "peer"
    # a run that cannot start is reported, not raised
    bad = bench.peer_leg(os.path.join(ROOT, "no_such_program"), "dam", 2, 1, 0, False)
    assert bad["status"].startswith("not started")


def test_one_gpu_references_are_keyed_by_window(tmp_path, monkeypatch):
    """The N > 1 line forms a speed-up only between two runs of the SAME window (round-4 verdict: one key `cfg4` held the
    developed-flow rate and was compared with the at-rest strong leg: 1.83x too high).  The cache is by leg, every entry carries
    (warm-up, steps per window, windows), an entry of another window — or a cache of the old layout — is not used, and
    leg_summary refuses to divide by a reference whose window differs from the leg's."""
    sys.path.insert(0, ROOT)
    import bench
    cache = tmp_path / "n1.json"
    monkeypatch.setattr(bench, "N1_CACHE", str(cache))
    assert bench.cached_n1() == {} and bench.one_gpu_reference(None, "cfg4_at_rest", 50, 200, 3, measure=False) is None
    bench.write_n1_cache({"cfg4_at_rest": bench.n1_leg(834.0, 50, 200, 3), "cfg4_developed": bench.n1_leg(455.5, 2000, 200, 3),
                          "cfg2_window": bench.n1_leg(12574.0, 5, 20, 1)})
    at_rest = bench.one_gpu_reference(None, "cfg4_at_rest", 50, 200, 3, measure=False)
    assert at_rest["timesteps_per_s"] == 834.0 and at_rest["reference"].startswith("cached") and at_rest["window"] == [50, 200, 3]
    assert bench.one_gpu_reference(None, "cfg4_developed", 2000, 200, 3, measure=False)["timesteps_per_s"] == 455.5
    assert bench.one_gpu_reference(None, "cfg4_developed", 50, 200, 3, measure=False) is None        # another window: not this entry
    assert bench.one_gpu_reference(None, "cfg4_at_rest", 50, 200, 1, measure=False) is None
    assert bench.one_gpu_reference(None, "cfg2_window", 200, 1000, 1, measure=False) is None
    # the legs' names are what the N = 1 run caches under
    assert {v[0] for v in bench.STRONG_LEGS.values()} == {"cfg4_at_rest", "cfg4_developed"}
    d = {"mparticle_steps_per_s": 1.0, "ticks_per_s": 2502.0, "ms_per_step": 0.4, "workload": "cfg4", "n_fluid": 32000000, "host": "c",
         "particles_conserved": True, "neighbour_rebuilds": 0, "warmup": 50, "steps": 200, "windows": 3}
    s = bench.leg_summary(d, 8, at_rest)
    assert s["speedup_vs_1gpu"] == 3.0 and s["one_gpu_timesteps_per_s"] == 834.0 and s["reference_window"] == s["window"] == [50, 200, 3]
    wrong = dict(at_rest, window=[2000, 200, 3])
    assert "speedup_vs_1gpu" not in bench.leg_summary(d, 8, wrong)
    assert "speedup_vs_1gpu" not in bench.leg_summary(d, 8, None)
    # a cache written by round 4's bench.py ({"cfg2": .., "cfg4": ..}: no schema) is ignored
    cache.write_text(json.dumps({"cfg2_window": 12574.0, "cfg2": 9924.0, "cfg4": 455.5}))
    assert bench.cached_n1() == {}


def test_a_leg_that_never_returns_is_killed_and_counts_as_failed(monkeypatch):
    """the multi-rank legs of the bench run under a time limit (bench.run_guarded): a transport that has never run between GPUs may
    also never return, and a bench that hangs reports nothing.  The whole process group goes (the launcher and its ranks)."""
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setenv("SPH_BENCH_LEG_TIMEOUT", "0.5")
    t0 = time.time()
    rc, so = bench.run_guarded(["bash", "-c", "echo started; sleep 30 & sleep 30"], dict(os.environ), "a leg that hangs")
    assert rc == 124 and time.time() - t0 < 10 and b"started" in so
    monkeypatch.setenv("SPH_BENCH_LEG_TIMEOUT", "30")
    rc, so = bench.run_guarded(["bash", "-c", "echo '{\"ok\": 1}'; exit 3"], dict(os.environ), "a leg that fails")
    assert rc == 3 and b'{"ok": 1}' in so


def test_the_faster_of_two_transports_that_agree_is_the_headline():
    """--transport best (opt-in; the default is the RCCL run, the peer run beside it): the RCCL run is the line's value unless the guarded peer run of the same window
    completed, computed the same flow (every particle owned once, max rho and max speed equal to rounding) and is faster."""
    sys.path.insert(0, ROOT)
    import bench
    rccl = {"particles_conserved": True, "n_fluid": 16000000, "n_gpus": 8, "steps": 20, "warmup": 5, "workload": "dam", "max_rho": 1003.21,
            "max_speed": 0.0412, "ticks_per_s": 7000.0}
    peer = dict(rccl, ticks_per_s=9100.0, max_rho=1003.2101, max_speed=0.04121)
    assert bench.transports_agree(rccl, peer)
    assert bench.choose_headline(rccl, {"status": "ok", "raw": peer})[0] == "peer"
    assert bench.choose_headline(rccl, {"status": "ok", "raw": dict(peer, ticks_per_s=7050.0)})[0] == "rccl"       # not faster by the margin
    assert bench.choose_headline(rccl, {"status": "timed out"})[0] == "rccl"
    assert bench.choose_headline(rccl, None)[0] == "rccl"
    for bad in (dict(peer, particles_conserved=False), dict(peer, max_rho=1010.0), dict(peer, max_speed=0.08), dict(peer, steps=40),
                dict(peer, n_gpus=4), {k: v for k, v in peer.items() if k != "max_rho"}):
        assert not bench.transports_agree(rccl, bad), bad
        which, why = bench.choose_headline(rccl, {"status": "ok", "raw": bad})
        assert which == "rccl" and "agree" in why


@pytest.mark.gpu
def test_six_ranks_rehearsal_weak_and_strong_legs(sph, tmp_path):
    """The N > 1 line as the driver's 8-GPU node will get it, rehearsed on the one GPU of this box: `bench.py --gpus 6` over
    the peer transport (six processes — the most this pool lets touch one GPU at a time; the multi-GPU node runs eight), the
    weak-scaling run as the line's value and, under scaling_detail, the STRONG leg: cfg4 (32 000 000 particles under the tilt
    trace) cut into the same six slabs, with the per-rank breakdown of a step (device, ranks seen, particles, begin / reduce /
    pack / exchange / end) in both.  Small step counts: this is a rehearsal of the code path, not a measurement."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    # The N = 1 run of this host first (the driver's command, without the CPU leg): it caches the one-GPU rates by leg and window.
    # The rehearsal's developed strong leg is cut short (300 + 3 x 40 steps): its window is then NOT the cached one.
    cache = str(tmp_path / "n1.json")
    env.update(SPH_BENCH_N1_CACHE=cache, SPH_BENCH_STRONG_DEVELOPED="300,40,3")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu"], capture_output=True,
                       timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    n1 = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][0])
    legs = json.load(open(cache))["legs"]
    also = {e["cache_key"]: e for e in n1["also"] if e.get("cache_key")}
    assert legs["cfg4_at_rest"] == {"timesteps_per_s": also["cfg4_at_rest"]["timesteps_per_s"], "warmup": 50, "steps": 200, "windows": 3}
    assert legs["cfg4_developed"]["timesteps_per_s"] == also["cfg4_developed"]["timesteps_per_s"] and legs["cfg4_developed"]["warmup"] == 2000
    assert legs["cfg4_at_rest"]["timesteps_per_s"] > 1.3 * legs["cfg4_developed"]["timesteps_per_s"]      # (two regimes: what the mix-up cost)
    assert legs["cfg2_window"]["timesteps_per_s"] == n1["timesteps_per_s"] and legs["cfg2_window"]["warmup"] == 5
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "6", "--transport", "peer", "--steps", "12", "--warmup", "4"],
                       capture_output=True, timeout=1100, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 6 and out["scaling"] == "weak" and out["particles_conserved"] is True and out["config"]["n_fluid"] == 12000000
    pr = out["per_rank"]
    assert len(pr) == 6 and [q["rank"] for q in pr] == list(range(6)) and all(q["ranks_seen"] == 6 for q in pr)
    assert sum(q["owned"] for q in pr) == 12000000 and out["breakdown_steps"] > 0
    assert all(q["begin_us"] > 0 and q["end_us"] > 0 and q["exchange_us"] > 0 for q in pr), pr
    weak = out["scaling_detail"]["weak"]
    assert weak["n_fluid"] == 12000000
    # the weak leg's window (12 steps after 4) is not the cached headline window (20 after 5): measured in this run
    assert weak["reference"].startswith("measured in this run") and weak["reference_window"] == [4, 12, 1]
    assert weak["weak_efficiency_vs_1gpu"] == pytest.approx(weak["timesteps_per_s"] / weak["one_gpu_timesteps_per_s"], abs=1e-4)
    strong = out["scaling_detail"]["strong"]
    for name, st in strong.items():
        assert st["n_fluid"] == 32000000 and st["particles_conserved"] is True and "tilt" in st["workload"], name
        assert len(st["per_rank"]) == 6 and sum(q["owned"] for q in st["per_rank"]) == 32000000
        assert st["reference_window"] == st["window"]
        assert st["speedup_vs_1gpu"] == pytest.approx(st["timesteps_per_s"] / st["one_gpu_timesteps_per_s"], abs=1e-3)
    # at rest: the window of the N = 1 run's cfg4 leg -> ITS cached figure (not the developed one: round 4's 1.83x)
    assert strong["at_rest"]["window"] == [50, 200, 3] and strong["at_rest"]["reference"].startswith("cached")
    assert strong["at_rest"]["one_gpu_timesteps_per_s"] == legs["cfg4_at_rest"]["timesteps_per_s"]
    # developed: this run's window differs from the cached one -> rank 0 measured the one-GPU rate itself, on that window
    assert strong["developed"]["window"] == [300, 40, 3] and strong["developed"]["reference"].startswith("measured in this run")
    assert strong["developed"]["one_gpu_timesteps_per_s"] != legs["cfg4_developed"]["timesteps_per_s"]


@pytest.mark.gpu
def test_box_calibration_is_plausible(sph):
    """sph_box_calibrate (include/sph_diag.h; the `box` key of the bench line): what THIS box delivers to a streaming copy and to a
    saturated v_fma_f32 stream — between a third of the HBM specification and the specification, between 2 and 4 SIMD-cycles per
    wave-instruction at the nominal clock, a shader clock between 1.2 and 2.6 GHz (MI355X: 2.4 GHz peak) — and repeatable to 10 %."""
    b = sph.box_calibrate(0, 3)
    assert 2500.0 < b["copy_gbs"] < 8000.0, b
    assert 1.9 < b["valu_cycles"] < 4.0, b
    assert b["clock_ghz"] is None or 1.2 < b["clock_ghz"] < 2.6, b
    assert max(b["copy_gbs_runs"]) < 1.1 * min(b["copy_gbs_runs"]) and max(b["valu_cycles_runs"]) < 1.1 * min(b["valu_cycles_runs"]), b

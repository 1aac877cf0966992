"""The C multi-GPU host (pi-sph-fluid_amd/host/slab_sph_fluid.c: one process per GPU, RCCL over xGMI) and the C
partitioner it uses (host/sph_slab_host.c).

CPU: the C partitioner equals the Python one (pi-sph-fluid_amd/slab.py, itself pinned by the gloo tests), a rank
generates exactly the lattice columns the Python host would give it, and `slab_sph_fluid --ranks 2` started without a
GPU fails loudly and at once (no CPU path, no rendezvous hang).
GPU (one MI355X): the one-rank run through ncclCommInitRank / the slab entry points equals sph_step on a single
context (--check), conserves every particle on the 2M-particle scene, and asking for more ranks than GPUs is refused
with a message instead of a hang."""
import ctypes as C
import json
import os
import subprocess
import time

import numpy as np
import pytest

from conftest import ROOT

HOST = os.path.join(ROOT, "pi-sph-fluid_amd", "host", "slab_sph_fluid")
HOST_STRESS = HOST + "_stress"      # `make stress`: the same host linked against the test build of the library (-DSPH_TEST_HOOKS)


def test_c_partitioner_equals_python(sph):
    L = sph.host_lib()
    for spec, worlds in ((sph.dam_break_spec(1), (1, 2, 3, 8)), (sph.BLOCK_SCENES["cfg4"], (1, 4, 8)),
                         (((0.0, 40.0, 0.0, 8.0), 0.3, 0.3, 240, 60), (1, 2, 5))):
        box, x0, y0, nx, ny = spec
        for skin in (0.15, 0.0, 0.3):
            prm = sph.default_params(box, skin)
            assert L.sph_slab_grid_columns(C.byref(prm)) == sph.slab.grid_columns(prm)
            for world in worlds:
                cuts = (C.c_int * (world + 1))()
                assert L.sph_slab_partition_block(C.byref(prm), x0, nx, ny, world, cuts) == 0
                py = sph.slab.partition_block(prm, spec, world)
                assert [(cuts[r], cuts[r + 1]) for r in range(world)] == py
                c0, c1 = py[world // 2]
                ib, ie = C.c_long(), C.c_long()
                assert L.sph_slab_block_columns(C.byref(prm), x0, nx, c0, c1, C.byref(ib), C.byref(ie)) == 0
                gc = sph.slab.block_lattice_columns(prm, spec)
                sel = np.nonzero((gc >= c0 - 2) & (gc < c1 + 2))[0]
                assert (ib.value, ie.value) == (int(sel[0]), int(sel[-1]) + 1)
    prm = sph.default_params((0.0, 4.0, 0.0, 2.0))
    cuts = (C.c_int * 9)()
    assert L.sph_slab_partition_block(C.byref(prm), 0.3, 20, 10, 8, cuts) == sph.SPH_E_ARG      # too narrow for 8 slabs


def test_c_host_fails_loudly_without_gpu(sph):
    if sph.hip_lib().sph_device_count() > 0:
        pytest.skip("a GPU is present: covered by the gpu tests below")
    t0 = time.time()
    r = subprocess.run([HOST, "--ranks", "2", "--block", "400", "100", "60", "20", "--steps", "3", "--warmup", "1"],
                       capture_output=True, timeout=120)
    assert r.returncode != 0
    assert b"no HIP device available" in r.stderr and b"no CPU path" in r.stderr
    assert not r.stdout.strip()
    assert time.time() - t0 < 60


@pytest.mark.gpu
@pytest.mark.parametrize("lean", ["auto", "0"])
def test_c_host_one_rank_equals_sph_step(sph, lean):
    """one slab through the C host against sph_step (--check): as the lean step (round 5: one call, four kernels — what a rank
    without neighbours and with the device to itself gets by default) and as the three-call step (--lean 0)"""
    r = subprocess.run([HOST, "--ranks", "1", "--block", "600", "150", "90", "20", "--steps", "150", "--warmup", "50", "--check", "--lean", lean],
                       capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    out = r.stdout.decode().splitlines()
    rec = json.loads([ln for ln in out if ln.startswith("{")][0])      # (RCCL prints a version banner on stdout first)
    assert rec["n_gpus"] == 1 and rec["n_fluid"] == 90000 and rec["particles_conserved"] is True
    assert rec["ticks_per_s"] > 0 and 0 < rec["neighbour_rebuilds"] < 200
    assert ("lean step" in rec["host"]) == (lean == "auto")
    chk = [ln for ln in out if ln.startswith("check:")]
    assert len(chk) == 1 and chk[0].endswith("-> ok"), chk


@pytest.mark.gpu
@pytest.mark.parametrize("order", ["serial", "main", "side"])
def test_c_host_rccl_calls_of_a_step_against_itself(sph, order):
    """--selfcomm: one rank issues the RCCL calls every step of an N > 1 run makes — the all-reduce on the device word, the
    grouped send / receive of both halo buffers — to itself, in each of the three stream orders (--exchange-stream): the
    streams, events and RCCL enqueues of the multi-GPU step on a one-GPU box.  What arrives is ignored (the slab has no
    neighbours), so the run must still equal sph_step (--check)."""
    r = subprocess.run([HOST, "--ranks", "1", "--block", "600", "150", "90", "20", "--steps", "150", "--warmup", "50", "--check",
                        "--selfcomm", "--exchange-stream", order], capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    out = r.stdout.decode().splitlines()
    rec = json.loads([ln for ln in out if ln.startswith("{")][0])
    assert rec["n_fluid"] == 90000 and rec["particles_conserved"] is True and 0 < rec["neighbour_rebuilds"] < 200
    chk = [ln for ln in out if ln.startswith("check:")]
    assert len(chk) == 1 and chk[0].endswith("-> ok"), chk


@pytest.mark.gpu
def test_c_host_tilt_run_and_rank_count_guard(sph):
    r = subprocess.run([HOST, "--ranks", "1", "--scene", "dam", "--steps", "60", "--warmup", "20", "--tilt"],
                       capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    rec = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][0])
    assert rec["n_fluid"] == 2000000 and rec["particles_conserved"] is True and "tilt" in rec["workload"]
    ndev = sph.hip_lib().sph_device_count()
    r = subprocess.run([HOST, "--ranks", str(ndev + 1), "--block", "400", "100", "60", "20", "--steps", "2", "--warmup", "1"],
                       capture_output=True, timeout=120)
    assert r.returncode != 0 and b"RCCL does not share a device" in r.stderr


@pytest.mark.gpu
def test_c_host_two_ranks_over_rccl(sph):
    """two ranks on two GPUs (skipped on a one-GPU box: RCCL does not share a device): the halo exchange and the reduction of
    the rebuild word over RCCL; every particle owned exactly once at the end, rebuilds in step on both ranks."""
    if sph.hip_lib().sph_device_count() < 2:
        pytest.skip("needs two GPUs")
    r = subprocess.run([HOST, "--ranks", "2", "--block", "600", "150", "90", "20", "--steps", "300", "--warmup", "100"],
                       capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    rec = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][0])
    assert rec["n_gpus"] == 2 and rec["n_fluid"] == 90000 and rec["particles_conserved"] is True
    assert rec["ticks_per_s"] > 0 and 0 < rec["neighbour_rebuilds"] < 400


@pytest.mark.gpu
def test_c_host_deterministic_equals_sph_step_bitwise(sph):
    """--deterministic (sph_params.deterministic): the slab path through the C host and sph_step give the same bits when
    they rebuild in the same steps (skin 0: every step; with a skin the slab path may rebuild a step earlier)"""
    r = subprocess.run([HOST, "--ranks", "1", "--block", "600", "150", "90", "20", "--steps", "150", "--warmup", "50", "--check",
                        "--deterministic", "--skin", "0"], capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    chk = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("check:")]
    assert len(chk) == 1 and "max|dx| = 0.000e+00" in chk[0] and "max|drho| = 0.000e+00" in chk[0], chk


def test_c_partition_counts_equals_python(sph):
    """the re-balancing cuts of the C host (sph_slab_partition_counts) equal the Python host's (_cuts_from_histogram)"""
    L = sph.host_lib()
    rng = np.random.default_rng(3)
    for cols, world in ((64, 2), (300, 3), (5352, 8), (40, 4)):
        hist = rng.integers(0, 500, cols).astype(np.int64)
        hist[: cols // 5] = 0                                     # a dry stretch
        cuts = (C.c_int * (world + 1))()
        assert L.sph_slab_partition_counts(hist.ctypes.data_as(C.POINTER(C.c_longlong)), cols, world, cuts) == 0
        assert [(cuts[r], cuts[r + 1]) for r in range(world)] == sph.slab._cuts_from_histogram(hist, world, 0, cols)
    hist = np.ones(10, np.int64)
    cuts = (C.c_int * 5)()
    assert L.sph_slab_partition_counts(hist.ctypes.data_as(C.POINTER(C.c_longlong)), 10, 4, cuts) == sph.SPH_E_ARG


def _run_host(args, timeout=600, env=None, host=HOST):
    r = subprocess.run([host] + [str(a) for a in args], capture_output=True, timeout=timeout, env=None if env is None else dict(os.environ, **env))
    out = r.stdout.decode().splitlines()
    rec = [json.loads(ln) for ln in out if ln.startswith("{")]
    return r, out, (rec[0] if rec else None)


def _flying_block(sph, u=5.0, deterministic=True, skin=0.0):
    prm, f, b = sph.scene_block((0.0, 90.0, 0.0, 20.0), 0.3, 0.3, 600, 150)
    prm.deterministic = 1 if deterministic else 0
    if skin is not None:
        prm.skin = prm.skin_min = skin
    f["u"] = u
    return prm, f, b


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [2, 4])
def test_c_host_ranks_over_peer_mapped_memory_equal_sph_step_bitwise(sph, tmp_path, ranks):
    """--transport peer on one GPU: the ranks (processes) map each other's receive buffers and flag words through hipIpc
    handles; per step sph_slab_peer_reduce (the rebuild word: MAX over the ranks through slot stores), sph_slab_peer_push (the
    halo buffers stored into the neighbours' memory, arrival flags) and sph_slab_peer_wait, all on the one stream.  Same
    scene and same bar as the shared-memory transport below: bit for bit sph_step's particles after 300 steps."""
    state = tmp_path / "state.bin"
    r, out, rec = _run_host(["--ranks", ranks, "--transport", "peer", "--block", 600, 150, 90, 20, "--velocity", 5, 0, "--steps", 250,
                             "--warmup", 50, "--deterministic", "--skin", 0, "--dump-state", state])
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    assert rec["n_gpus"] == ranks and rec["n_fluid"] == 90000 and rec["particles_conserved"] is True and rec["neighbour_rebuilds"] >= 300
    assert "peer-mapped" in rec["host"]
    got = np.fromfile(state, sph.PARTICLE)
    prm, f, b = _flying_block(sph)
    with sph.Context(prm, f, b, 0.0, -9.81) as ctx:
        ctx.step(300, 0.0, -9.81)
        ctx.sync()
        ref = ctx.read_particles()
    assert len(got) == len(ref)
    for k in ("x", "y", "u", "v", "rho", "p"):
        assert np.array_equal(got[k], ref[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [2, 4])
def test_c_host_lean_step_over_peer_mapped_memory_equals_sph_step_bitwise(sph, tmp_path, ranks):
    """The LEAN step (sph_slab_step: head | ghost update or rebuild | density | force — four kernels, nothing of the exchange as a
    launch of its own) with real neighbours, on one GPU: the ranks cap the grids of their one-launch kernels (--one-launch-wgs)
    so that all of them stay resident.  Skin 0: every step rebuilds — the pack as the first phase of the rebuild launch, the
    records pushed into the neighbours' mapped receive buffers and theirs awaited between two of its grid barriers, the rebuild
    word exchanged by the head kernel.  Bit for bit sph_step's particles after 300 steps of a block flying through the
    interfaces."""
    prm, f, b = _flying_block(sph)
    with sph.Context(prm, f, b, 0.0, -9.81) as ctx:
        ctx.step(300, 0.0, -9.81)
        ctx.sync()
        ref = ctx.read_particles()
    for spec in (2, 1, 0):      # (round 6: the fused speculative lean step (3 launches), the speculative one (the word goes round in the gate kernel), the plain one)
        state = tmp_path / ("state%d.bin" % spec)
        r, out, rec = _run_host(["--ranks", ranks, "--transport", "peer", "--lean", 1, "--lean-spec", spec, "--one-launch-wgs", 256, "--block", 600, 150, 90, 20,
                                 "--velocity", 5, 0, "--steps", 250, "--warmup", 50, "--deterministic", "--skin", 0, "--dump-state", state])
        assert r.returncode == 0, r.stderr.decode()[-3000:]
        assert rec["n_gpus"] == ranks and rec["n_fluid"] == 90000 and rec["particles_conserved"] is True and rec["neighbour_rebuilds"] >= 300
        assert "peer-mapped" in rec["host"] and "lean step" in rec["host"] and ("speculative" in rec["host"]) == bool(spec) and ("fused" in rec["host"]) == (spec == 2)
        got = np.fromfile(state, sph.PARTICLE)
        assert len(got) == len(ref)
        for k in ("x", "y", "u", "v", "rho", "p"):
            assert np.array_equal(got[k], ref[k]), (k, spec)


@pytest.mark.gpu
def test_c_host_lean_step_updates_equal_the_three_call_step_bitwise(sph, tmp_path):
    """... and with the default (adaptive) skin, where most steps carry UPDATES: the message the force pass of the step before left
    in the send buffers, pushed by the head kernel into the parity buffer of the step, awaited by the ghost-update launch itself.
    Three ranks, 400 steps, deterministic order: the same bits as the three-call step over the same transport (which packs
    nothing either any more, but pushes and waits with kernels of its own) — and re-balancing on top (contexts re-created, the
    peer blocks and their flags kept) ends cleanly with every particle owned once."""
    rebuilds = {}
    for verify in (1, 0):      # (1: failing boxes verified particle by particle by blocks of the head kernel — opt-in for slabs: sph_set_verification)
        states = []
        for lean in (1, 0):
            state = tmp_path / ("state%d%d.bin" % (lean, verify))
            r, out, rec = _run_host(["--ranks", 3, "--transport", "peer", "--lean", lean, "--lean-spec", 0, "--one-launch-wgs", 256, "--block", 600, 150, 90, 20,
                                     "--velocity", 5, 0, "--steps", 300, "--warmup", 100, "--deterministic", "--verify", verify, "--dump-state", state])
            assert r.returncode == 0, r.stderr.decode()[-3000:]
            assert rec["particles_conserved"] is True and 0 < rec["neighbour_rebuilds"] < 200 and ("lean step" in rec["host"]) == bool(lean)
            states.append((np.fromfile(state, sph.PARTICLE), rec["neighbour_rebuilds"]))
        assert states[0][1] == states[1][1]                                  # the ranks rebuilt in the same steps
        for k in ("x", "y", "u", "v", "rho", "p"):
            assert np.array_equal(states[0][0][k], states[1][0][k]), (k, verify)
        rebuilds[verify] = states[0][1]
    assert rebuilds[1] <= rebuilds[0]                                        # verification only ever saves rebuilds
    # re-balancing under the lean step: the block of test_c_host_rebalancing_..., flying along x at 30 m/s out of the first slab and
    # into the last — the contexts are re-created (their step counts, the tags of the lean step's messages, start from 1 again:
    # the flags and word slots of the peer blocks with them)
    r, out, rec = _run_host(["--ranks", 3, "--transport", "peer", "--lean", 1, "--lean-spec", 0, "--one-launch-wgs", 256, "--block", 160, 40, 60, 6,
                             "--origin", 2.0, 1.5, "--velocity", 30, 0, "--capacity", 3200, "--warmup", 0, "--steps", 900, "--rebalance-every", 150])
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    assert rec["particles_conserved"] is True and rec["rebalanced"] >= 3 and "lean step" in rec["host"]
    assert rec["max_owned"] <= 6400 // 3 + 700


@pytest.mark.gpu
def test_c_host_lean_step_with_a_rank_held_up_between_its_launches(sph, tmp_path):
    """Ranks that share a device are time-sliced: a rank's launches can be held up for longer than its neighbour needs for a whole
    step.  In the lean step the neighbour is then a launch AHEAD — the push blocks of its next head kernel do not wait for that
    launch's exchange block — and the arrival flag reads 2 (t + 1) where the rank that was held up waits for 2 t.  (Round 5: the
    wait was for equality; the four-rank bitwise test above gave up in one run of many with 'a neighbouring rank's ... did not
    arrive'.)  $SPH_TEST_STALL_AFTER_HEAD holds rank 1's host for 300 us between the head kernel and the rest of every third step:
    the run ends cleanly and with the bits of the run nobody held up — the message of step t is still in the buffer of its parity.
    (Round 6: the hook is compiled into the TEST build of the library only — `make stress`, -DSPH_TEST_HOOKS — and both runs go through
    host/slab_sph_fluid_stress, the C host linked against it; sph_slab_step one call per step: --lean-graph 0; the four-launch speculative
    form, --lean-spec 1: the fused form has no host between a step's launches to hold up.)"""
    if not os.path.exists(HOST_STRESS):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "pi-sph-fluid_amd"), "stress"])
    states = []
    for stall in (None, "1:300:3"):
        state = tmp_path / ("state_%s.bin" % (stall is not None))
        r, out, rec = _run_host(["--ranks", 3, "--transport", "peer", "--lean", 1, "--one-launch-wgs", 256, "--block", 600, 150, 90, 20,
                                 "--velocity", 5, 0, "--steps", 200, "--warmup", 40, "--deterministic", "--lean-graph", 0, "--lean-spec", 1, "--dump-state", state],
                                env=None if stall is None else {"SPH_TEST_STALL_AFTER_HEAD": stall}, host=HOST_STRESS)
        assert r.returncode == 0, r.stderr.decode()[-3000:]
        assert rec["particles_conserved"] is True and 0 < rec["neighbour_rebuilds"] < 200 and "lean step" in rec["host"]
        states.append((np.fromfile(state, sph.PARTICLE), rec["neighbour_rebuilds"], rec["ticks_per_s"]))
    assert states[0][1] == states[1][1]
    assert states[1][2] < 0.9 * states[0][2]                       # (the hook did hold the run up)
    for k in ("x", "y", "u", "v", "rho", "p"):
        assert np.array_equal(states[0][0][k], states[1][0][k]), k


@pytest.mark.gpu
def test_c_host_peer_transport_default_skin_and_rebalancing(sph, tmp_path):
    """the peer transport with the default (adaptive) skin — most steps carry updates, the ranks must agree on the steps
    that rebuild — and re-balancing (contexts re-created, the peer blocks kept): particles conserved, the run ends cleanly"""
    r, out, rec = _run_host(["--ranks", 3, "--transport", "peer", "--block", 600, 150, 90, 20, "--velocity", 5, 0, "--steps", 500,
                             "--warmup", 100, "--rebalance-every", 200])
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    assert rec["particles_conserved"] is True and 0 < rec["neighbour_rebuilds"] < 600


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [2, 4])
def test_c_host_ranks_over_shared_memory_equal_sph_step_bitwise(sph, tmp_path, ranks):
    """THE STEP LOOP OF THE C HOST WITH NEIGHBOURS, on one GPU: 2 and 4 ranks (processes) exchange the halo buffers and
    reduce the rebuild word through POSIX shared memory (--transport host: sph_slab_copy_out / _copy_in, sph_slab_flag_get /
    _set) in the order the RCCL transport uses.  A block flying along x at 5 m/s (particles migrate between the slabs),
    deterministic order, skin 0 (both sides rebuild every step): after 300 steps the ranks' particles equal sph_step's bit by bit."""
    state = tmp_path / "state.bin"
    r, out, rec = _run_host(["--ranks", ranks, "--transport", "host", "--block", 600, 150, 90, 20, "--velocity", 5, 0, "--steps", 250,
                             "--warmup", 50, "--deterministic", "--skin", 0, "--dump-state", state])
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    assert rec["n_gpus"] == ranks and rec["n_fluid"] == 90000 and rec["particles_conserved"] is True and rec["neighbour_rebuilds"] >= 300
    assert "shared memory" in rec["host"]
    got = np.fromfile(state, sph.PARTICLE)
    prm, f, b = _flying_block(sph)
    with sph.Context(prm, f, b, 0.0, -9.81) as ctx:
        ctx.step(300, 0.0, -9.81)
        ctx.sync()
        ref = ctx.read_particles()
    assert len(got) == len(ref)
    for k in ("x", "y", "u", "v", "rho", "p"):
        assert np.array_equal(got[k], ref[k]), k
    # the slabs did exchange owners: the block moved 0.37 m = 1.6 cell columns
    assert np.abs(ref["x"] - f["x"]).max() > 0.3


@pytest.mark.gpu
def test_c_host_shared_memory_default_skin_console_and_frame(sph, tmp_path):
    """default skin (lists reused; the ranks must agree on the steps that rebuild): every particle owned once, the run close to
    sph_step's; the reference's console line (:679-691) with the maxima reduced over the ranks; the metaball frame OR-ed
    from the ranks' pages equals the single context's up to threshold pixels (a 8 x 4 m box: pixels of 6 cm)."""
    state, frame = tmp_path / "state.bin", tmp_path / "frame.bin"
    r, out, rec = _run_host(["--ranks", 3, "--transport", "host", "--block", 80, 40, 8, 4, "--velocity", 1, 0, "--steps", 500,
                             "--warmup", 0, "--console", "--frame", frame, "--dump-state", state])
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    assert rec["particles_conserved"] is True and 0 < rec["neighbour_rebuilds"] < 500
    assert out[0].startswith("dt = ") and out[1] == "n_fluid = 3200" and out[2].startswith("n_boundary = ")
    lines = [ln for ln in out if ln.startswith("sim time: ")]
    assert len(lines) == 1                                        # 500 steps = 0.122 s of simulated time: one line
    import re
    m = re.match(r"sim time: (\d+\.\d\d), ticks/s: (\d+), max rho error: (-?\d+\.\d{3})% \(worst\) (-?\d+\.\d{3})%, "
                 r"max speed: (\d+\.\d) m/s \(worst\) (\d+\.\d) m/s, $", lines[0])
    assert m, lines[0]
    got = np.fromfile(state, sph.PARTICLE)
    prm, f, b = sph.scene_block((0.0, 8.0, 0.0, 4.0), 0.3, 0.3, 80, 40)
    f["u"] = 1.0
    with sph.Context(prm, f, b, 0.0, -9.81) as ctx:
        ctx.step(500, 0.0, -9.81)
        ctx.sync()
        ref = ctx.read_particles()
        page = ctx.render_metaballs()
        mr, ms = ctx.stats()
    # (a lattice block collapsing under gravity: the two runs differ by summation order and by the steps in which they
    # rebuild, and the flow amplifies that; the bit-exact comparison is the test above)
    assert max(np.abs(got["x"] - ref["x"]).max(), np.abs(got["y"] - ref["y"]).max()) <= 5e-3
    assert np.max(np.abs(got["rho"] - ref["rho"]) / ref["rho"]) <= 2e-2
    assert abs(rec["max_rho"] - mr) <= 1e-2 * mr and abs(rec["max_speed"] - ms) <= 2e-2 * ms
    assert 0.5 < float(m.group(5)) < 4.0                          # the block moves at 1 m/s and falls
    fr = np.fromfile(frame, np.uint8)
    assert len(fr) == 1024 and fr.any() and page.any()
    assert int(np.unpackbits(fr ^ page).sum()) <= 4


@pytest.mark.gpu
def test_c_host_rebalancing_keeps_a_migrating_flow_inside_capacity(sph, tmp_path):
    """Re-balancing in the C host: a block flying along x at 30 m/s leaves the first slab and piles into the
    last one.  With the static partition the last rank runs out of particle capacity (SPH_E_CAPACITY: reported, and the
    launcher ends the other ranks); with --rebalance-every 150 the run goes through, every particle stays owned exactly once,
    the slabs stay balanced and the result is the single context's (a re-created slab re-evaluates a from (x, v): the run is
    continued, not bit-continued; the block moves as a whole, so the comparison is tight)."""
    base = ["--ranks", 3, "--transport", "host", "--block", 160, 40, 60, 6, "--origin", 2.0, 1.5, "--velocity", 30, 0,
            "--capacity", 3200, "--warmup", 0, "--steps", 900]
    r, out, rec = _run_host(base)
    assert r.returncode != 0 and b"capacity" in r.stderr, r.stderr.decode()[-2000:]
    state = tmp_path / "state.bin"
    r, out, rec = _run_host(base + ["--rebalance-every", 150, "--dump-state", state])
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    assert rec["particles_conserved"] is True and rec["rebalanced"] >= 3
    # (balanced at step 750; in the 150 steps since, 1.1 m = 5 cell columns of 120 particles have flown on)
    assert rec["max_owned"] <= 6400 // 3 + 700
    got = np.fromfile(state, sph.PARTICLE)
    prm, f, b = sph.scene_block((0.0, 60.0, 0.0, 6.0), 2.0, 1.5, 160, 40)
    f["u"] = 30.0
    with sph.Context(prm, f, b, 0.0, -9.81) as ctx:
        ctx.step(900, 0.0, -9.81)
        ctx.sync()
        ref = ctx.read_particles()
    assert max(np.abs(got["x"] - ref["x"]).max(), np.abs(got["y"] - ref["y"]).max()) <= 5e-4
    assert np.abs(ref["x"] - f["x"]).min() > 6.0                  # the block did fly: 30 m/s x 0.22 s


@pytest.mark.gpu
@pytest.mark.parametrize("spec", [0, 1, 2])
def test_c_host_graphed_lean_steps_equal_single_calls_bitwise(sph, tmp_path, spec):
    """sph_slab_steps (round 6): runs of 16 / 8 / 4 / 2 lean steps replayed as captured graphs — step number, buffer parity and gravity
    taken from device memory — against sph_slab_step, one call of four launches per step: three ranks over the peer transport on one
    GPU, the adaptive skin (most steps carry updates, some rebuild), the tilt trace (gravity changes between runs of steps),
    deterministic order: the same rebuild steps and the same bits.  1 + 163 + 120 steps: runs that are no multiple of anything.
    spec = 1: the speculative lean step (sph_slab_set_speculative) the same way."""
    states = []
    for graph in (1, 0):
        state = tmp_path / ("state_g%d.bin" % graph)
        r, out, rec = _run_host(["--ranks", 3, "--transport", "peer", "--lean", 1, "--lean-graph", graph, "--lean-spec", spec, "--one-launch-wgs", 256, "--block", 600, 150, 90, 20,
                                 "--velocity", 5, 0, "--steps", 163, "--warmup", 121, "--tilt", "--deterministic", "--dump-state", state])
        assert r.returncode == 0, r.stderr.decode()[-3000:]
        assert rec["particles_conserved"] is True and 0 < rec["neighbour_rebuilds"] < 200
        assert ("graphs of up to 16 steps" in rec["host"]) == bool(graph) and ("speculative" in rec["host"]) == bool(spec)
        states.append((np.fromfile(state, sph.PARTICLE), rec["neighbour_rebuilds"]))
    assert states[0][1] == states[1][1]
    for k in ("x", "y", "u", "v", "rho", "p"):
        assert np.array_equal(states[0][0][k], states[1][0][k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("spec", [0, 1, 2])
def test_c_host_lean_step_state_and_last_step_vs_oracle(sph, orc, oracle, tmp_path, spec):
    """The lean step against the ORACLE (round 5's tests of it were bitwise against sph_step / the three-call step: transitive).  Two
    ranks over the peer transport, graphed runs of steps, the block flying through the interface at 5 m/s with the default (adaptive)
    skin: the C host dumps the gathered state and accelerations in front of the last step and behind it, and ONE oracle step from the
    former must give the latter (conftest.fused_step_vs_oracle: x, v_half, rho, p, a, v and the acceleration the kick used: all
    within 1e-5 on their scales) — the integration of a slab step, the ghosts' densities included, pinned to pi_sph_fluid.c:612-641.
    spec = 1: the speculative lean step (the criterion inside the density launch, the word exchanged by the gate kernel); 2: its fused
    form (the head's work by the first workgroups of the density launch, ghost-staging tiles wait for them: three launches)."""
    from conftest import fused_step_vs_oracle
    fn = {k: tmp_path / (k + ".bin") for k in ("s0", "a0", "s1", "a1")}
    r, out, rec = _run_host(["--ranks", 2, "--transport", "peer", "--lean", 1, "--lean-spec", spec, "--one-launch-wgs", 256, "--block", 600, 150, 90, 20, "--velocity", 5, 0,
                             "--steps", 150, "--warmup", 50, "--dump-before", fn["s0"], fn["a0"], "--dump-state", fn["s1"], "--dump-accel", fn["a1"]])
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    assert rec["particles_conserved"] is True and 0 < rec["neighbour_rebuilds"] < 200 and "graphs of up to 16 steps" in rec["host"]
    before, after = np.fromfile(fn["s0"], sph.PARTICLE), np.fromfile(fn["s1"], sph.PARTICLE)
    a0, a1 = np.fromfile(fn["a0"], np.float32).reshape(-1, 2), np.fromfile(fn["a1"], np.float32).reshape(-1, 2)
    prm, f, b = _flying_block(sph, deterministic=False, skin=None)
    assert len(before) == len(after) == len(f) == 90000
    assert np.abs(before["x"] - f["x"]).min() > 0.1        # (the block has flown: 199 steps at 5 m/s = 0.24 m)
    p = oracle.params((prm.x_min, prm.x_max, prm.y_min, prm.y_max))
    ob = b.view(orc.PARTICLE).copy()
    oracle.psi(p, ob)
    worst = fused_step_vs_oracle(orc, oracle, p, ob, before, (a0[:, 0].copy(), a0[:, 1].copy()), after, (a1[:, 0].copy(), a1[:, 1].copy()),
                                 (0.0, -9.81), float(np.float32(prm.dt)), tag="lean step, 2 ranks, peer")
    assert max(worst.values()) <= 1.0, worst


@pytest.mark.gpu
def test_c_host_speculative_lean_step(sph, tmp_path):
    """The speculative lean step (round 6, sph_slab_set_speculative; --lean-spec 1): one rank against sph_step (--check) — and what it
    is for: on the dam break a slab through the plain lean step rebuilds its neighbour structure more than twice as often as sph_step
    (failing boxes ask for the rebuild: the verification would sit on the step's critical path), through the speculative one about as
    often as sph_step (the verification rides in the density launch).  Three ranks over the peer transport with re-balancing: every
    particle owned once."""
    for mode in (1, 2):
        r = subprocess.run([HOST, "--ranks", "1", "--block", "600", "150", "90", "20", "--steps", "150", "--warmup", "50", "--check", "--lean-spec", str(mode)],
                           capture_output=True, timeout=600)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        out = r.stdout.decode().splitlines()
        rec = json.loads([ln for ln in out if ln.startswith("{")][0])
        assert rec["particles_conserved"] is True and "speculative" in rec["host"] and ("fused" in rec["host"]) == (mode == 2)
        chk = [ln for ln in out if ln.startswith("check:")]
        assert len(chk) == 1 and chk[0].endswith("-> ok"), chk
    rebuilds = {}
    for spec in (2, 1, 0):
        r, out, rec = _run_host(["--ranks", 1, "--scene", "dam", "--steps", 400, "--warmup", 1200, "--lean-spec", spec, "--verify", 1 if spec else -1])
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        assert rec["particles_conserved"] is True
        rebuilds[spec] = rec["neighbour_rebuilds"]
    assert rebuilds[1] < 0.7 * rebuilds[0] and rebuilds[2] < 0.7 * rebuilds[0], rebuilds
    for mode in (1, 2):
        r, out, rec = _run_host(["--ranks", 3, "--transport", "peer", "--lean", 1, "--lean-spec", mode, "--one-launch-wgs", 256, "--block", 160, 40, 60, 6,
                                 "--origin", 2.0, 1.5, "--velocity", 30, 0, "--capacity", 3200, "--warmup", 0, "--steps", 900, "--rebalance-every", 150])
        assert r.returncode == 0, r.stderr.decode()[-3000:]
        assert rec["particles_conserved"] is True and rec["rebalanced"] >= 3 and "speculative" in rec["host"]

"""Manual (not collected): steps/s of cfg2 over the windows of SURVEY 8(d) (200 warm-up, 5 x 1000) and of the developed flow
(steps 4000-9000) for a few skins.  Usage: python tools/skin_sweep_gpu.py 0.10 0.15 0.12:0.30 ...  (a:b = adaptive
between a and b; one number = fixed; "@name" appended: with csrc/libsph_hip_name.so, a `make variant` build)"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sph = importlib.import_module("pi-sph-fluid_amd")
DEFAULT_LIB = sph.LIB_HIP
for spec in sys.argv[1:] or ["0.12:0.30"]:
    spec, _, variant = spec.partition("@")
    sph.LIB_HIP = DEFAULT_LIB.replace("libsph_hip.so", "libsph_hip_%s.so" % variant) if variant else DEFAULT_LIB
    sph._hip = None
    prm, f, b = sph.dam_break(1)
    lo, hi = [float(x) for x in (spec.split(":") if ":" in spec else (spec, spec))]
    prm.skin_min, prm.skin = lo, hi
    ctx = sph.Context(prm, f, b)
    ctx.step(200); ctx.sync()
    rates, rebs, skins = [], [], []
    for w in range(9):
        r0 = ctx.rebuild_stats()[0]
        t0 = time.perf_counter(); ctx.step(1000); ctx.sync(); dt = time.perf_counter() - t0
        rates.append(1000 / dt); rebs.append((ctx.rebuild_stats()[0] - r0) / 1000); skins.append(ctx.current_skin())
    d = ctx.time_kernel("density_eos", 30) * 1e3; fo = ctx.time_kernel("force_kick", 30) * 1e3
    print("skin %s%s: windows %s | rebuilds/step %s | skin at window ends %s | median first five %.0f, last five (4200-9200) %.0f | density %.1f force %.1f us (developed)" % (
        spec, "@" + variant if variant else "", " ".join("%.0f" % r for r in rates), " ".join("%.3f" % r for r in rebs), " ".join("%.3f" % r for r in skins), np.median(rates[:5]), np.median(rates[4:]), d, fo), flush=True)
    del ctx

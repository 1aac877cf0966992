"""Manual (not collected): density / force / list-build / whole-rebuild timings of one or more builds of the library on
cfg2 after a warm-up.  Usage: python tools/kb2_gpu.py [warmup] lib1.so lib2.so ..."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sph = importlib.import_module("pi-sph-fluid_amd")
warm = int(sys.argv[1]) if len(sys.argv) > 1 else 300
libs = sys.argv[2:] or [sph.LIB_HIP]
for rnd in range(2):
    for lib in libs:
        sph.LIB_HIP = lib if os.path.isabs(lib) else os.path.join(ROOT, lib)
        sph._hip = None
        prm, f, b = sph.dam_break(1)
        ctx = sph.Context(prm, f, b)
        ctx.step(warm); ctx.sync()
        d = ctx.time_kernel("density_eos", 50) * 1e3
        fo = ctx.time_kernel("force_kick", 50) * 1e3
        bl = ctx.time_kernel("build_list", 20) * 1e3
        ctx.step(1); ctx.sync()
        import time
        ctx.step(200); ctx.sync()
        t0 = time.perf_counter(); ctx.step(1000); ctx.sync(); t1 = time.perf_counter()
        r0 = ctx.rebuild_stats()
        acc = {}
        for _ in range(10):
            ctx.request_rebuild()
            kt = ctx.profile_steps(1)
            for k, v in kt.items(): acc[k] = acc.get(k, 0.0) + v / 10
        print("%-22s density %.2f force %.2f build_list %.2f us | 1000 steps: %.0f steps/s (rebuilds, direct %s) | rebuild-every-step step %.1f us" % (
            os.path.basename(lib), d, fo, bl, 1000 / (t1 - t0), r0, acc["step"] * 1e3), flush=True)
        ctx.close() if hasattr(ctx, "close") else None
        del ctx

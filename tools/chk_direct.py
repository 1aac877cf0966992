import importlib, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sph = importlib.import_module("pi-sph-fluid_amd")
prm, f, b = sph.dam_break(1)
with sph.Context(prm, f, b, 0.0, -9.81) as ctx:
    ctx.step(50, 0.0, -9.81); ctx.sync()
    print("rebuild stats", ctx.rebuild_stats(), "reasons", ctx.direct_tile_reasons() if hasattr(ctx, "direct_tile_reasons") else None)
    print("dens", ctx.time_kernel("density_eos", 20) * 1e3, "force", ctx.time_kernel("force_kick", 20) * 1e3)

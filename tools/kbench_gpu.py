"""Manual kernel timing (not collected by pytest): cfg2 after warm-up, back-to-back launches of the two heavy
kernels, then the force-kernel ablation builds (SPH_ABLATE bits: 1 = gathers read one address, 2 = no pair
arithmetic, 4 = no staging) when the measurement library exists (`make -C pi-sph-fluid_amd ablate`: the only build that
reads $SPH_ABLATE).  Usage: python tools/kbench_gpu.py [warmup_steps] [skin]"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sph = importlib.import_module("pi-sph-fluid_amd")
warm = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ablate_lib = os.path.join(ROOT, "pi-sph-fluid_amd", "csrc", "libsph_hip_ablate.so")
if os.path.exists(ablate_lib):
    sph.LIB_HIP = ablate_lib
prm, f, b = sph.dam_break(1)
if len(sys.argv) > 2:
    prm.skin = prm.skin_min = float(sys.argv[2])
os.environ.pop("SPH_ABLATE", None)
ctx = sph.Context(prm, f, b)
ctx.step(warm)
ctx.sync()
print("lib %s skin %.3f rebuilds/direct tiles %s" % (os.path.basename(sph.LIB_HIP), prm.skin, ctx.rebuild_stats()))
for abl in (("0", "1", "2", "3", "0") if sph.LIB_HIP == ablate_lib else ("0",)):
    os.environ["SPH_ABLATE_DENS"] = abl
    print("density SPH_ABLATE_DENS=%s : %.2f us" % (abl, ctx.time_kernel("density_eos", 50) * 1e3))
os.environ.pop("SPH_ABLATE_DENS", None)
for abl in (("0", "1", "2", "4", "7", "8", "15", "16", "31", "32", "63", "0") if sph.LIB_HIP == ablate_lib else ("0",)):
    os.environ["SPH_ABLATE"] = abl
    print("force SPH_ABLATE=%s : %.2f us" % (abl, ctx.time_kernel("force_kick", 50) * 1e3))
os.environ.pop("SPH_ABLATE", None)
# what a rebuild costs: a rebuild requested before every step, per-kernel times by HIP events
acc = {}
for _ in range(10):
    ctx.request_rebuild()
    kt = ctx.profile_steps(1)
    for k, v in kt.items():
        acc[k] = acc.get(k, 0.0) + v / 10
print("rebuild every step:", {k: round(v * 1e3, 1) for k, v in acc.items() if k not in ("rebuilds_per_step",)}, "rebuilds/step", acc["rebuilds_per_step"])
# the list build alone, back to back on the live sort (SPH_ABLATE_BUILD: 1 = no walk, 2 = no staging, 4 = lists not
# written, 8 = stop after the runs);
# whatever it leaves is replaced by the rebuild of the next step
for abl in (("0", "1", "2", "3", "8", "0") if sph.LIB_HIP == ablate_lib else ("0",)):
    os.environ["SPH_ABLATE_BUILD"] = abl
    print("build_list SPH_ABLATE_BUILD=%s : %.2f us" % (abl, ctx.time_kernel("build_list", 20) * 1e3))
if sph.LIB_HIP == ablate_lib:      # clock stamps of the phases of a few tiles (printed by the kernel)
    os.environ["SPH_ABLATE_BUILD"] = "64"
    ctx.time_kernel("build_list", 1)
    ctx.sync()
    ctx.set_rebuild_launches(True)   # ... and of the barriers of a one-launch rebuild (drops the captured graphs)
    ctx.step(2)
    ctx.sync()
    ctx.request_rebuild()
    ctx.step(1)
    ctx.sync()
os.environ.pop("SPH_ABLATE_BUILD", None)
ctx.step(1)
ctx.sync()
print("after:", ctx.rebuild_stats())

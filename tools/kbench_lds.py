"""tools/kbench_lds.py [warmup] — how much of the force pass is LDS bank conflicts?  (GPU box; `make -C pi-sph-fluid_amd ablate`.)
The force kernel without its epilogue (SPH_ABLATE=16) against the same with every gather at a conflict-free address (144:
consecutive lanes read consecutive slots; same instructions, same rows), and that without the staging of the tile (148: the
upper bound of anything that makes the tile fill cheaper — LDS-DMA, a prefetch under the previous tile's walk), alternating, on
cfg2 after `warmup` steps."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sph = importlib.import_module("pi-sph-fluid_amd")
sph.LIB_HIP = os.path.join(ROOT, "pi-sph-fluid_amd", "csrc", "libsph_hip_ablate.so")
warm = int(sys.argv[1]) if len(sys.argv) > 1 else 300
prm, f, b = sph.dam_break(1)
os.environ.pop("SPH_ABLATE", None)
with sph.Context(prm, f, b) as ctx:
    ctx.step(warm)
    ctx.sync()
    for abl in ("0", "16", "144", "148", "16", "144", "148", "16", "144", "148", "0"):
        os.environ["SPH_ABLATE"] = abl
        print("force SPH_ABLATE=%s : %.2f us" % (abl, ctx.time_kernel("force_kick", 100) * 1e3))

"""tools/listlen_stats.py [warmup] — how much of the list walkers' work is padding to a wave's longest list, and how much of it a
permutation of the lanes INSIDE a tile by list length would remove.  A cfg2 state after `warmup` steps (default 4000) is read back;
the neighbour counts within 2H (core entries) and within 2H + skin (all entries) are recomputed on the host (scipy cKDTree), the
particles put in the tile order (column pair, row, column: 256 consecutive = a tile, 64 = a wave) and the rows of four entries a
wave walks — the maximum over its lanes — summed: as built, and with the lanes of each tile sorted by count.  (GPU box.)"""
import importlib, os, sys
import numpy as np
from scipy.spatial import cKDTree
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sph = importlib.import_module("pi-sph-fluid_amd")
warm = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
prm, f, b = sph.dam_break(1)
with sph.Context(prm, f, b, 0.0, -9.81) as ctx:
    ctx.step(warm, 0.0, -9.81)
    ctx.sync()
    p = ctx.read_particles()
    skin = ctx.current_skin()
    rows, cols, cell = ctx.device_grid()
x, y = p["x"].astype(np.float64), p["y"].astype(np.float64)
H2 = 2.0 * float(prm.h)
tree = cKDTree(np.stack([x, y], 1))
core = tree.query_ball_point(np.stack([x, y], 1), H2, return_length=True, workers=-1) - 1
full = tree.query_ball_point(np.stack([x, y], 1), H2 * (1.0 + skin), return_length=True, workers=-1) - 1
row = np.clip((y / cell).astype(np.int64), 0, rows - 1)
col = np.clip((x / cell).astype(np.int64), 0, cols - 1)
order = np.lexsort((col & 1, row, col >> 1))          # tile order: pair, row, column inside the pair
n = len(x) // 256 * 256
for name, cnt in (("core (d < 2H: what the force pass walks at least)", core), ("all entries (d < 2H + skin: the density pass)", full)):
    c = cnt[order][:n].reshape(-1, 4, 64)             # tile, wave, lane
    r4 = lambda a: (a + 3) // 4
    as_built = r4(c.max(axis=2)).sum()
    srt = np.sort(cnt[order][:n].reshape(-1, 256), axis=1).reshape(-1, 4, 64)
    sorted_ = r4(srt.max(axis=2)).sum()
    ideal = r4(c).sum() / 64.0                        # every lane only its own rows
    print("%s: skin %.3f, mean count %.1f; rows per wave: as built %.2f, lanes sorted inside the tile %.2f (%.1f %% fewer), no padding at all %.2f"
          % (name, skin, cnt.mean(), as_built / c.shape[0] / 4, sorted_ / c.shape[0] / 4, 100.0 * (1 - sorted_ / as_built), ideal / c.shape[0] / 4))

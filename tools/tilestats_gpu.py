"""Manual (not collected): distribution of the list kernels' per-workgroup staging sizes (TILE_CAP = 1280) on a live
cfg2 state, recomputed on the host from a read-back."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sph = importlib.import_module("pi-sph-fluid_amd")
warm = int(sys.argv[1]) if len(sys.argv) > 1 else 300
prm, f, b = sph.scene(sys.argv[2]) if len(sys.argv) > 2 else sph.dam_break(1)
ctx = sph.Context(prm, f, b)
ctx.step(warm); ctx.sync()
p = ctx.read_particles()
rows, cols, cell = ctx.device_grid()          # the device's own grid: cell = 2H + skin
inv = np.float32(1.0) / np.float32(cell)
row = np.minimum(np.maximum((p["y"] * inv).astype(np.int64), 0), rows - 1)
col = np.minimum(np.maximum((p["x"] * inv).astype(np.int64), 0), cols - 1)
key = np.sort(col * rows + row)
n = len(key); ncell = rows * cols
cs = np.searchsorted(key, np.arange(ncell + 1))
nb = (n + 255) // 256
klo = key[np.arange(nb) * 256]; khi = key[np.minimum(np.arange(nb) * 256 + 255, n - 1)]
tot = np.zeros(nb, np.int64)
HALF = 127
for s in range(3):
    lo = klo + (s - 1) * rows - 1; hi = khi + (s - 1) * rows + 1
    ok = (hi >= 0) & (lo <= ncell - 1)
    lo = np.clip(lo, 0, ncell - 1); hi = np.clip(hi, 0, ncell - 1)
    tot += np.where(ok, cs[hi + 1] - cs[lo], 0)
span = khi - klo
print("blocks", nb, "total: mean %.0f p50 %d p90 %d p99 %d max %d ; >1152: %.1f%%  >1280: %.1f%%" % (
    tot.mean(), np.percentile(tot, 50), np.percentile(tot, 90), np.percentile(tot, 99), tot.max(), 100 * np.mean(tot > 1152), 100 * np.mean(tot > 1280)))
print("tiles above", {t: int((tot > t).sum()) for t in (896, 960, 1024, 1056, 1088, 1120, 1152)})
print("span: mean %.0f p50 %d p90 %d p99 %d max %d ; >250: %.1f%%" % (span.mean(), np.percentile(span, 50), np.percentile(span, 90), np.percentile(span, 99), span.max(), 100 * np.mean(span > 250)))
cnt = np.diff(cs); occ = cnt[cnt > 0]
print("particles per occupied cell: mean %.2f max %d ; per 3-cell column range max %d" % (occ.mean(), occ.max(), np.convolve(cnt, np.ones(3, int), "same").max()))

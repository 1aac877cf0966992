"""tools/spec_trace.py [warmup [steps]] — where does the time of the speculative density launch go?  (GPU box; needs the
trace build: make -C pi-sph-fluid_amd variant NAME=trace VFLAGS=-DSPH_SPEC_TRACE.)  cfg2, `warmup` steps, then `steps` steps one
at a time: per step the begin / end clocks of every workgroup of the launch (check jobs, tiles, verify jobs) and the verify jobs'
counters, printed per age of the lists (steps since the last rebuild)."""
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sph = importlib.import_module("pi-sph-fluid_amd")
sph.LIB_HIP = os.path.join(ROOT, "pi-sph-fluid_amd", "csrc", os.environ.get("SPH_TRACE_LIB", "libsph_hip_trace.so"))
warmup = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
prm, f, b = sph.dam_break(1)
n = len(f)
L = sph.hip_lib()
L.sph_spec_trace.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
L.sph_spec_trace.restype = C.c_int
tiles = (n + 255) // 256
tiles_grid = (tiles + 63) // 64 * 64
nw = (n + 63) // 64
ncheck = min(256, max(8, ((nw * 8 + 255) // 256 + 7) & ~7))      # spec_check_wgs (sph_list.inc): CHECK_LANES = 8
nverify = min(int(os.environ.get("SPH_TRACE_VERIFY_WGS", "512")), max(8, (nw // 4 + 7) & ~7))      # spec_verify_wgs (SPH_SPEC_VERIFY_WGS)
tiles_a = (tiles_grid * int(os.environ.get("SPH_TRACE_AT", "5")) // 8) & ~7
grid = ncheck + tiles_grid + nverify
buf = np.zeros((grid, 2), dtype=np.uint64)
cnt = np.zeros(4, dtype=np.uint32)
rows = []
with sph.Context(prm, f, b, 0.0, -9.81, device=0) as ctx:
    ctx.step(warmup, 0.0, -9.81)
    ctx.sync()
    L.sph_spec_trace(buf.ctypes.data, grid, cnt.ctypes.data)
    age = 0
    for s in range(steps):
        r0 = ctx.rebuild_stats()[0]
        v0 = ctx.verify_stats()
        ctx.step(1, 0.0, -9.81)
        ctx.sync()
        rebuilt = ctx.rebuild_stats()[0] != r0
        L.sph_spec_trace(buf.ctypes.data, grid, cnt.ctypes.data)
        t = buf.astype(np.int64)
        live = t[:, 1] > 0
        t0 = t[:, 0][live].min()
        us = (t - t0) / 100.0      # 100 MHz
        ck, tl_a = us[:ncheck], us[ncheck:ncheck + tiles_a]
        vf, tl_b = us[ncheck + tiles_a:ncheck + tiles_a + nverify], us[ncheck + tiles_a + nverify:]
        tl_b = tl_b[:tiles - tiles_a]
        rows.append((age, int(rebuilt), ck[:, 1].max(), vf[:, 0].min(), vf[:, 1].max(), max(tl_a[:, 1].max(), tl_b[:, 1].max()),
                     us[live][:, 1].max(), int(ctx.verify_stats() - v0), int(cnt[0]), int(cnt[1]), int(cnt[2]),
                     float(np.median(vf[:, 1] - vf[:, 0])), float((vf[:, 1] - vf[:, 0]).max()), float(np.median(ck[:, 1] - ck[:, 0]))))
        age = 0 if rebuilt else age + 1
print("age reb | check_end verify_begin verify_end tiles_end launch_end | queued taken past_filter trips | verify_wg med max  check_wg med")
for r in rows:
    print("%3d %d | %6.1f %6.1f %6.1f %6.1f %6.1f | %5d %5d %5d %6d | %5.1f %5.1f  %5.1f" % r)
a = np.array([r[2:7] for r in rows if not r[1]])
print("mean (steps without a rebuild): check_end %.1f verify_begin %.1f verify_end %.1f tiles_end %.1f launch_end %.1f" % tuple(a.mean(axis=0)))

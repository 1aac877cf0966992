# tools/ab_quiet_drv.sh : the driver's window (20 steps after 5) eight times, alternating: skin_min 0.12 (round 4's floor), the default
# (0.08 at rest: the quiet floor), and a fixed skin of 0.15 — one library
for rep in 1 2 3 4 5 6 7 8; do
for v in "floor_0.12 SPH_BENCH_SKIN_MIN=0.12" "default _X=0" ; do
set -- $v
env $2 python bench.py --no-cpu --no-also --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
print('$1', d['timesteps_per_s'], d['kernel_ms']['force_kick']*1e3, d['kernel_ms']['density_eos']*1e3)"
done; done | python -c "
import sys, collections, statistics
v=collections.defaultdict(list)
for ln in sys.stdin:
    a,b,c,d=ln.split(); v[a].append((float(b),float(c),float(d)))
for k,x in v.items(): print('%-12s steps/s mean %8.1f median %8.1f min %8.1f max %8.1f | force %.1f us density %.1f us (n=%d)' % (k, statistics.mean(t[0] for t in x), statistics.median(t[0] for t in x), min(t[0] for t in x), max(t[0] for t in x), statistics.mean(t[1] for t in x), statistics.mean(t[2] for t in x), len(x)))"

# tools/ab_drv.sh lib... : the driver's window (bench.py --steps 20 --warmup 5) eight times per build, alternating; mean and spread
for rep in 1 2 3 4 5 6 7 8; do
for lib in "$@"; do
python bench.py --no-cpu --no-also --lib pi-sph-fluid_amd/csrc/$lib --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
print('$lib', d['timesteps_per_s'])"
done; done | python -c "
import sys, collections, statistics
v=collections.defaultdict(list)
for ln in sys.stdin:
    a,b=ln.split(); v[a].append(float(b))
for k,x in v.items(): print('%-24s mean %8.1f  median %8.1f  min %8.1f  max %8.1f  (n=%d)' % (k, statistics.mean(x), statistics.median(x), min(x), max(x), len(x)))"

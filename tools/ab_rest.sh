# tools/ab_rest.sh lib... : the driver's N = 1 command (20 steps after 5: the fluid still at rest) and 200 steps after 5, three times per build
for rep in 1 2 3; do
for lib in "$@"; do
for k in 20 200; do
python bench.py --no-cpu --no-also --lib pi-sph-fluid_amd/csrc/$lib --steps $k --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
print('$lib', 'steps $k: %.0f' % d['timesteps_per_s'], 'dens %.1f force %.1f' % (d['kernel_ms']['density_eos']*1e3, d['kernel_ms']['force_kick']*1e3), d['neighbour_rebuilds_per_step'])"
done; done; done

# tools/ab_dev.sh lib... : cfg2 steps/s in the developed flow (1000 steps after 4000) and the early window (after 200), twice per build
for rep in 1 2; do
for lib in "$@"; do
for w in 4000 200; do
python bench.py --no-cpu --no-also --lib pi-sph-fluid_amd/csrc/$lib --steps 1000 --warmup $w 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
print('$lib', 'w$w %.0f' % d['timesteps_per_s'], 'dens %.1f force %.1f' % (d['kernel_ms']['density_eos']*1e3, d['kernel_ms']['force_kick']*1e3), d['neighbour_rebuilds_per_step'], d['rebuild_requests'])"
done; done; done

# tools/ab_small.sh lib... : the small scenes (cfg0: 269 particles, cfg1: 100 000) steps/s per build of libsph_hip
for lib in "$@"; do
for wl in cfg0 cfg1; do
for w in 200 2200; do
python bench.py --no-cpu --no-also --workload $wl --lib pi-sph-fluid_amd/csrc/$lib --steps 2000 --warmup $w 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
print('$lib', '$wl w$w %.0f' % d['timesteps_per_s'], 'dens %.1f force %.1f' % (d['kernel_ms']['density_eos']*1e3, d['kernel_ms']['force_kick']*1e3), d['neighbour_rebuilds_per_step'], d['rebuild_requests'])"
done; done; done

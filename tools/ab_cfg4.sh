# tools/ab_cfg4.sh lib... : cfg4 (32 M particles, tilt trace) steps/s in the developed flow (600 steps after 2000) per build
for lib in "$@"; do
python bench.py --no-cpu --no-also --workload cfg4 --tilt --lib pi-sph-fluid_amd/csrc/$lib --steps 600 --warmup 2000 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
print('$lib', 'cfg4 w2000 %.0f' % d['timesteps_per_s'], 'dens %.1f force %.1f' % (d['kernel_ms']['density_eos']*1e3, d['kernel_ms']['force_kick']*1e3), d['neighbour_rebuilds_per_step'], d['rebuild_requests'])"
done

# tools/skin_max_sweep.sh : cfg2 steps/s after 200 / 1200 / 4000 steps with the adaptive skin's upper end at 0.30 (default) / 0.35 / 0.40 / 0.45
for mx in 0.30 0.35 0.40 0.45; do
for w in 200 1200 4000; do
SPH_BENCH_SKIN_MAX=$mx python bench.py --no-cpu --no-also --steps 1000 --warmup $w 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
print('skin_max $mx', 'w$w %.0f' % d['timesteps_per_s'], 'dens %.1f force %.1f' % (d['kernel_ms']['density_eos']*1e3, d['kernel_ms']['force_kick']*1e3), d['neighbour_rebuilds_per_step'], d['rebuild_requests'], 'skin at end %.3f' % d['skin_at_end_fraction_of_2h'])"
done; done

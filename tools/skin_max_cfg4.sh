# tools/skin_max_cfg4.sh : cfg4's developed flow (600 steps after 2000, tilt trace) with the adaptive skin's upper end at 0.30 / 0.40 / 0.50
for mx in 0.30 0.40 0.50; do
SPH_BENCH_SKIN_MAX=$mx python bench.py --no-cpu --no-also --workload cfg4 --tilt --steps 600 --warmup 2000 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
print('skin_max $mx', 'cfg4 w2000 %.0f' % d['timesteps_per_s'], 'dens %.1f force %.1f' % (d['kernel_ms']['density_eos']*1e3, d['kernel_ms']['force_kick']*1e3), d['neighbour_rebuilds_per_step'], d['rebuild_requests'], 'skin at end %.3f' % d['skin_at_end_fraction_of_2h'])"
done

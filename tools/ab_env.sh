# tools/ab_env.sh VAR : cfg2 steps/s in three windows with and without the environment variable VAR set (same library, same box)
for rep in 1 2; do for v in "" "$@"; do
for w in 200 1200 4000; do
env ${v:+$v=1} python bench.py --no-cpu --no-also --steps 1000 --warmup $w 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
print('${v:-default}', 'w$w %.0f' % d['timesteps_per_s'], 'dens %.1f force %.1f' % (d['kernel_ms']['density_eos']*1e3, d['kernel_ms']['force_kick']*1e3), d['neighbour_rebuilds_per_step'])"
done; done; done

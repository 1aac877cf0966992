# tools/skin_sweep.sh [warmup] : cfg2 steps/s over 1000 steps after `warmup` (default 4000) with the adaptive skin and with fixed skins
W=${1:-4000}
for sk in "" 0.10 0.12 0.14 0.16 0.18 0.21 0.25; do
python bench.py --no-cpu --no-also --steps 1000 --warmup $W ${sk:+--skin $sk} 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
print('skin ${sk:-adaptive}', 'w$W %.0f' % d['timesteps_per_s'], 'dens %.1f force %.1f' % (d['kernel_ms']['density_eos']*1e3, d['kernel_ms']['force_kick']*1e3), d['neighbour_rebuilds_per_step'], d['rebuild_requests'], 'skin at end %.3f' % d['skin_at_end_fraction_of_2h'])"
done

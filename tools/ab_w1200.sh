# tools/ab_w1200.sh [VAR=VALUE ...] : cfg2's collapse window (1000 steps after 1200) with each environment setting, twice; prints the step rate,
# the speculative density launch (with its criterion jobs) next to the plain pass, the verified group pairs per step, requests and repairs
for rep in 1 2; do
for e in "_X=0" "$@"; do
env $e python bench.py --no-cpu --no-also --warmup 1200 --steps 1000 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
k=d['kernel_ms']
print('%-28s %9.1f steps/s  dens plain %.1f spec+reset %.1f force %.1f  rebuilds/step %.4f verified/step %.0f skin end %.3f' % ('$e', d['timesteps_per_s'], k['density_eos']*1e3, k.get('density_spec_launch_plus_reset',0)*1e3, k['force_kick']*1e3, d['neighbour_rebuilds_per_step'], d['verified_group_pairs_per_step'], d['skin_at_end_fraction_of_2h']), d['rebuild_requests'], d.get('list_repairs'))"
done; done

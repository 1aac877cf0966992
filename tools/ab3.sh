for lib in "$@"; do
python bench.py --no-cpu --no-also --lib pi-sph-fluid_amd/csrc/$lib --steps 1000 --warmup 200 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
print('$lib', 'cfg2 %.0f' % d['timesteps_per_s'], d['neighbour_rebuilds_per_step'], d['rebuild_requests'], d['verified_group_pairs_per_step'])"
python bench.py --no-cpu --no-also --lib pi-sph-fluid_amd/csrc/$lib --steps 1000 --warmup 1200 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
print('$lib', 'cfg2 w1200 %.0f' % d['timesteps_per_s'], d['neighbour_rebuilds_per_step'], d['rebuild_requests'], d['verified_group_pairs_per_step'])"
done

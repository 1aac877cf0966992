# tools/ab_tune.sh lib... : cfg2 after 200 / 1200 / 4000 steps (1000 each) and the 8(d) protocol's windows in one run, per build, twice
run() {
python bench.py --no-cpu --no-also --lib pi-sph-fluid_amd/csrc/$1 --workload cfg2 --warmup $2 --steps 1000 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
print('%-22s w%-5s %9.1f steps/s  rebuilds/step %.4f verified/step %.0f skin end %.3f' % ('$1', '$2', d['timesteps_per_s'], d['neighbour_rebuilds_per_step'], d['verified_group_pairs_per_step'], d['skin_at_end_fraction_of_2h']), d['rebuild_requests'])"
}
for rep in 1 2; do
for lib in "$@"; do
run $lib 200; run $lib 1200; run $lib 2200; run $lib 4000
done; done

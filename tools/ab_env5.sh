# tools/ab_env5.sh VAR=VALUE : the in-tree build with and without an environment setting — cfg2 after 200 / 1200 / 4000 steps (1000 steps
# each), cfg4's developed flow (600 after 2000, tilt), cfg1; twice, alternating.  e.g. tools/ab_env5.sh SPH_NO_LIST_REPAIR=1
run() {
env $1 python bench.py --no-cpu --no-also --workload $2 --warmup $3 --steps $4 ${@:5} 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
k=d['kernel_ms']
print('%-24s %-5s w%-5s k%-5s %9.1f steps/s  dens %.1f force %.1f  rebuilds/step %.4f' % ('$1', '$2', '$3', '$4', d['timesteps_per_s'], k['density_eos']*1e3, k['force_kick']*1e3, d['neighbour_rebuilds_per_step']), d['rebuild_requests'], d.get('list_repairs'))"
}
for rep in 1 2; do
for e in "_X=0" "$1"; do
run $e cfg2 200 1000; run $e cfg2 1200 1000; run $e cfg2 4000 1000; run $e cfg4 2000 600 --tilt; run $e cfg1 5500 1500
done; done

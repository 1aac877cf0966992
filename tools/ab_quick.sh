# tools/ab_quick.sh lib... : cfg2 developed (1000 after 4000), cfg2 after 200, cfg4 developed (600 after 2000), alternating builds, twice
run() {
python bench.py --no-cpu --no-also --lib pi-sph-fluid_amd/csrc/$1 --workload $2 --warmup $3 --steps $4 ${@:5} 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
k=d['kernel_ms']
print('%-22s %-5s w%-5s k%-5s %9.1f steps/s  dens %.1f force %.1f  rebuilds/step %.4f' % ('$1', '$2', '$3', '$4', d['timesteps_per_s'], k['density_eos']*1e3, k['force_kick']*1e3, d['neighbour_rebuilds_per_step']), d['rebuild_requests'])"
}
for rep in 1 2; do
for lib in "$@"; do
run $lib cfg2 200 1000; run $lib cfg2 4000 1000; run $lib cfg4 2000 600 --tilt
done; done

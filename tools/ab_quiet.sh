# tools/ab_quiet.sh : the quiet floor of the skin controller (round 5: skin_min 0.08, reached only after lists that lived >= 100 steps)
# against the library before it (libsph_hip_r5a.so with skin_min 0.12: floor 0.12 always) — ab5's regimes, twice, alternating
run() {  # label lib env workload warmup steps extra...
env $3 python bench.py --no-cpu --no-also --lib pi-sph-fluid_amd/csrc/$2 --workload $4 --warmup $5 --steps $6 ${@:7} 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
k=d['kernel_ms']
print('%-10s %-5s w%-5s k%-5s %9.1f steps/s  dens %.1f force %.1f  rebuilds/step %.4f skin now %.3f' % ('$1', '$4', '$5', '$6', d['timesteps_per_s'], k['density_eos']*1e3, k['force_kick']*1e3, d['neighbour_rebuilds_per_step'], d.get('skin_at_end_fraction_of_2h', -1)), d['rebuild_requests'])"
}
for rep in 1 2; do
for v in "before libsph_hip_r5a.so SPH_BENCH_SKIN_MIN=0.12" "quiet libsph_hip.so _X=0"; do
set -- $v
run $1 $2 $3 cfg2 5 20; run $1 $2 $3 cfg2 5 200; run $1 $2 $3 cfg2 200 1000; run $1 $2 $3 cfg2 1200 1000; run $1 $2 $3 cfg2 4000 1000
run $1 $2 $3 cfg1 200 2000; run $1 $2 $3 cfg4 50 200 --tilt; run $1 $2 $3 cfg4 2000 600 --tilt
done; done

# tools/ab_rebuild_cost.sh [lib...] : what the step rate makes of the PRICE of a rebuild — builds whose rebuild costs N us more
# (make variant NAME=slowN VFLAGS=-DSPH_REBUILD_EXTRA_US=N) against the in-tree build, cfg2 after 200 / 1200 / 4000 steps (1000 steps each),
# twice, alternating: steps/s, rebuilds per step, the skin the controller ends on, the two walkers.  The slope d(us per step) / d(us per
# rebuild) is what a CHEAPER rebuild would return.  (GPU box.)
libs="${@:-libsph_hip.so libsph_hip_slow60.so libsph_hip_slow120.so}"
run() {  # lib warmup steps
python bench.py --no-cpu --no-also --lib pi-sph-fluid_amd/csrc/$1 --workload cfg2 --warmup $2 --steps $3 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
k=d['kernel_ms']
print('%-24s w%-5s k%-5s %9.1f steps/s %8.2f us/step  rebuilds/step %.4f  skin at end %.3f  dens %.1f force %.1f' % ('$1', '$2', '$3', d['timesteps_per_s'], 1e6/d['timesteps_per_s'], d['neighbour_rebuilds_per_step'], d['skin_at_end_fraction_of_2h'], k['density_eos']*1e3, k['force_kick']*1e3), d['rebuild_requests'])"
}
for rep in 1 2; do
for lib in $libs; do
run $lib 200 1000; run $lib 1200 1000; run $lib 4000 1000
done; done

# tools/ab_libs.sh "<workload warmup steps [extra]>;..." lib... : builds of libsph_hip against each other over the given windows, twice,
# alternating — steps/s, the two walkers (back-to-back launches), rebuilds per step.  e.g.
#   tools/ab_libs.sh "cfg2 5 200;cfg2 200 1000;cfg2 4000 1000" libsph_hip.so libsph_hip_rev.so
wins="$1"; shift
run() {  # lib workload warmup steps extra...
python bench.py --no-cpu --no-also --lib pi-sph-fluid_amd/csrc/$1 --workload $2 --warmup $3 --steps $4 ${@:5} 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
k=d['kernel_ms']
print('%-24s %-5s w%-5s k%-5s %9.1f steps/s %8.2f us/step  dens %.1f (spec launch %.1f) force %.1f  rebuilds/step %.4f' % ('$1', '$2', '$3', '$4', d['timesteps_per_s'], 1e6/d['timesteps_per_s'], k['density_eos']*1e3, k.get('density_spec_launch_plus_reset', 0)*1e3, k['force_kick']*1e3, d['neighbour_rebuilds_per_step']))"
}
for rep in 1 2; do
for lib in "$@"; do
IFS=';' read -ra W <<< "$wins"
for w in "${W[@]}"; do run $lib $w; done
done; done

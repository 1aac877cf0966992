run() {
python bench.py --no-cpu --no-also --lib pi-sph-fluid_amd/csrc/$1 --workload cfg4 --tilt --warmup $2 --steps $3 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
print('%-22s w%-5s k%-5s %9.1f steps/s rebuilds/step %.4f verified/step %.0f' % ('$1', '$2', '$3', d['timesteps_per_s'], d['neighbour_rebuilds_per_step'], d['verified_group_pairs_per_step']), d['rebuild_requests'], d.get('list_repairs'))"
}
for rep in 1 2; do for lib in libsph_hip.so libsph_hip_vq32k.so; do run $lib 50 600; run $lib 2000 600; done; done

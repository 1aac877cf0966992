# tools/occ_sweep.sh : the two walkers at fewer resident workgroups per compute unit (dynamic LDS that nobody uses: $SPH_DENS_EXTRA_LDS,
# $SPH_FORCE_EXTRA_LDS) — cfg2, 200 steps after 5 (at rest) and 1000 after 4000 (developed): steps/s, the speculative density launch and
# the force launch timed back to back.  Density: 8 / 7 / 6 / 5 / 4 / 3 per CU; force: 7 / 6 / 5 / 4 / 3.
# (the two environment hooks exist in the measurement build only: make -C pi-sph-fluid_amd ablate)
make -s -C pi-sph-fluid_amd ablate
run() {  # label env warmup steps
env $2 python bench.py --lib pi-sph-fluid_amd/csrc/libsph_hip_ablate.so --no-cpu --no-also --warmup $3 --steps $4 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); k=d['kernel_ms']
print('%-34s w%-5s %9.1f steps/s  density (speculative launch) %.1f us  plain %.1f us  force %.1f us' % ('$1', '$3', d['timesteps_per_s'], k.get('density_spec_launch_plus_reset',0)*1e3, k['density_eos']*1e3, k['force_kick']*1e3))"
}
for w in "5 200" "4000 1000"; do set -- $w
for e in "density_8_per_CU:_X=0" "density_7:SPH_DENS_EXTRA_LDS=13000" "density_6:SPH_DENS_EXTRA_LDS=16000" "density_5:SPH_DENS_EXTRA_LDS=20000" "density_4:SPH_DENS_EXTRA_LDS=26000" "density_3:SPH_DENS_EXTRA_LDS=34000" \
         "force_7_per_CU:_X=0" "force_6:SPH_FORCE_EXTRA_LDS=2000" "force_5:SPH_FORCE_EXTRA_LDS=6000" "force_4:SPH_FORCE_EXTRA_LDS=14000" "force_3:SPH_FORCE_EXTRA_LDS=24000"; do
run ${e%%:*} ${e##*:} $1 $2
done; done

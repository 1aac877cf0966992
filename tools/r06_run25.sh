#!/bin/bash
# round 6, GPU call 25: the fused speculative lean step at the slab sizes of the multi-GPU runs, two and four ranks on ONE GPU (large halos:
# hundreds of update blocks at the head of the density launch, ghost-staging tiles waiting for them)
cd "$(dirname "$0")/.." && . tools/gpu_steps.sh
H=pi-sph-fluid_amd/host/slab_sph_fluid
n=0
for cfg in "2 dam 1500 100" "4 dam 800 100" "2 cfg4 300 50 --tilt" "4 cfg4 300 50 --tilt" "4 cfg3 600 100"; do
  set -- $cfg; n=$((n+1))
  step 400 gpurun_out/r06_big_$n.txt $H --ranks $1 --transport peer --lean 1 --one-launch-wgs $((1024 / $1)) --scene $2 --steps $3 --warmup $4 $5
  grep -o '"host": "[^"]*"\|"ticks_per_s": [0-9.]*\|"particles_conserved": [a-z]*\|"neighbour_rebuilds": [0-9]*\|"n_fluid": [0-9]*' gpurun_out/r06_big_$n.txt | tr '\n' ' '; echo
done

#!/bin/bash
# round 6: soak of the speculative lean step over the peer transport with several ranks on ONE GPU (time-sliced: launches of one rank are
# held up for longer than a neighbour's whole step) — long runs (re-balanced: the block flies 7 m, a static partition runs out of capacity), with and without graphs, re-balancing, the tilt trace, the stall hook
cd "$(dirname "$0")/.." && . tools/gpu_steps.sh
H=pi-sph-fluid_amd/host/slab_sph_fluid
HS=pi-sph-fluid_amd/host/slab_sph_fluid_stress
B="--transport peer --lean 1 --one-launch-wgs 256 --block 600 150 90 20 --velocity 5 0"
n=0
for ranks in 2 3 4; do for g in 1 0; do for sp in ${SOAK_SPEC:-2 1 0}; do
  n=$((n+1))
  step 200 gpurun_out/r06_soak_$n.txt $H --ranks $ranks $B --lean-graph $g --lean-spec $sp --steps 6000 --warmup 100 --tilt --rebalance-every 1000
  grep -o '"ticks_per_s": [0-9.]*\|"particles_conserved": [a-z]*\|"neighbour_rebuilds": [0-9]*' gpurun_out/r06_soak_$n.txt | tr '\n' ' '; echo " <- ranks $ranks graph $g spec $sp"
done; done; done
for rep in 1 2 3; do
  n=$((n+1))
  step 300 gpurun_out/r06_soak_$n.txt $H --ranks 4 $B --lean-spec ${SOAK_LONG_SPEC:-2} --steps 12000 --warmup 100 --rebalance-every 1500
  grep -o '"ticks_per_s": [0-9.]*\|"particles_conserved": [a-z]*\|"rebalanced": [0-9]*' gpurun_out/r06_soak_$n.txt | tr '\n' ' '; echo " <- 4 ranks, re-balancing, rep $rep"
done
for stall in 1:300:3 2:800:5 0:150:2; do
  n=$((n+1))
  SPH_TEST_STALL_AFTER_HEAD=$stall step 300 gpurun_out/r06_soak_$n.txt $HS --ranks 4 $B --lean-graph 0 --lean-spec 1 --steps 3000 --warmup 100
  grep -o '"ticks_per_s": [0-9.]*\|"particles_conserved": [a-z]*' gpurun_out/r06_soak_$n.txt | tr '\n' ' '; echo " <- 4 ranks, stall $stall"
done

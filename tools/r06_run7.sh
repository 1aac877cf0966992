#!/bin/bash
# round 6, GPU call 7: the speculative lean step (relaxed polls), the whole GPU suite, slab repair scenes
cd "$(dirname "$0")/.." && . tools/gpu_steps.sh
H=pi-sph-fluid_amd/host/slab_sph_fluid
for sp in 1 0 1 0; do
step 120 gpurun_out/r06_spec_2M_$sp.json $H --ranks 1 --scene dam --steps 1000 --warmup 200 --windows 5 --lean-spec $sp
done
for sp in 1 0; do
step 200 gpurun_out/r06_spec_4Mdev_$sp.json $H --ranks 1 --scene cfg4slab --tilt --steps 200 --warmup 2000 --windows 3 --lean-spec $sp
step 200 gpurun_out/r06_spec_4Mrest_$sp.json $H --ranks 1 --scene cfg4slab --tilt --steps 200 --warmup 50 --windows 3 --lean-spec $sp
done
step 300 gpurun_out/r06_slab_repair_explore2.txt python tools/slab_repair_explore.py
step 1000 gpurun_out/r06_t_all.log python -m pytest tests -x -q -m gpu

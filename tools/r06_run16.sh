#!/bin/bash
# round 6, GPU call 16: the fused speculative lean step (3 launches) — tests and one slab through the C host
cd "$(dirname "$0")/.." && . tools/gpu_steps.sh
H=pi-sph-fluid_amd/host/slab_sph_fluid
step 200 gpurun_out/r06_fuse_check.txt $H --ranks 1 --block 600 150 90 20 --steps 150 --warmup 50 --check --lean-spec 2
step 200 gpurun_out/r06_fuse_2r.txt $H --ranks 2 --transport peer --lean 1 --lean-spec 2 --one-launch-wgs 256 --block 600 150 90 20 --velocity 5 0 --steps 300 --warmup 100
for sp in 2 1 2 1; do
step 120 gpurun_out/r06_fuse_2M_$sp.json $H --ranks 1 --scene dam --steps 1000 --warmup 200 --windows 5 --lean-spec $sp
done
for sp in 2 1; do
step 200 gpurun_out/r06_fuse_4Mdev_$sp.json $H --ranks 1 --scene cfg4slab --tilt --steps 200 --warmup 2000 --windows 3 --lean-spec $sp
step 200 gpurun_out/r06_fuse_4Mrest_$sp.json $H --ranks 1 --scene cfg4slab --tilt --steps 200 --warmup 50 --windows 3 --lean-spec $sp
done
step 900 gpurun_out/r06_t_slab_c3.log python -m pytest tests/test_slab_c_host.py -x -q -m gpu

#!/bin/bash
# round 6, GPU call 2: the graphed lean step (tests + one slab through the C host, graphs on / off), the price of a rebuild (A/B)
cd "$(dirname "$0")/.." && . tools/gpu_steps.sh
H=pi-sph-fluid_amd/host/slab_sph_fluid
step 900 gpurun_out/r06_t_slab_c.log python -m pytest tests/test_slab_c_host.py -x -q -m gpu
step 600 gpurun_out/r06_t_slab.log python -m pytest tests/test_gpu_slab.py -x -q -m gpu
for g in 1 0 1 0; do
step 120 gpurun_out/r06_slab1_2M_graph$g.json $H --ranks 1 --scene dam --steps 1000 --warmup 200 --windows 5 --lean-graph $g
done
for g in 1 0; do
step 200 gpurun_out/r06_slab1_4M_graph$g.json $H --ranks 1 --scene cfg4slab --tilt --steps 200 --warmup 50 --windows 3 --lean-graph $g
done
step 200 gpurun_out/r06_b2.json python bench.py --no-cpu --no-also --steps 1000 --warmup 200
step 600 gpurun_out/r06_ab_rebuild_cost.txt bash tools/ab_rebuild_cost.sh
step 300 gpurun_out/r06_ab_ratio3.txt bash tools/ab_rebuild_cost.sh libsph_hip.so libsph_hip_ratio3.so

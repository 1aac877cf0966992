#!/bin/bash
# round 6, GPU call 23: what the driver runs at round end — smoke(), the GPU suite with -x, the bench at N = 1
cd "$(dirname "$0")/.." && . tools/gpu_steps.sh
step 300 gpurun_out/r06_smoke.log python -c "import __graft_entry__ as g; g.smoke()"
step 1100 gpurun_out/r06_t_driver.log python -m pytest tests/ -x -q -m gpu
step 600 gpurun_out/r06_b10.json python bench.py --gpus 1 --steps 20 --warmup 5

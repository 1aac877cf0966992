#!/bin/bash
# round 6, GPU call 18: the fused speculative lean step as the C host's default — the suite, the full bench line, the slab profiles
cd "$(dirname "$0")/.." && . tools/gpu_steps.sh
step 1100 gpurun_out/r06_t_all3.log python -m pytest tests -q -m gpu
step 900 gpurun_out/r06_b6.json python bench.py
step 200 gpurun_out/r06_b7.json python bench.py --steps 20 --warmup 5 --no-also --no-cpu

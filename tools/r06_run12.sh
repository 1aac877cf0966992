#!/bin/bash
# round 6, GPU call 12: the full bench line (CPU baseline included) and the GPU suite on the final library
cd "$(dirname "$0")/.." && . tools/gpu_steps.sh
step 900 gpurun_out/r06_b4.json python bench.py
step 200 gpurun_out/r06_b5.json python bench.py --steps 20 --warmup 5 --no-also --no-cpu
step 1100 gpurun_out/r06_t_all2.log python -m pytest tests -q -m gpu

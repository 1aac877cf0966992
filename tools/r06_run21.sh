#!/bin/bash
# round 6, GPU call 21: the final library — the whole GPU suite, the full bench line, the driver's command
cd "$(dirname "$0")/.." && . tools/gpu_steps.sh
step 1100 gpurun_out/r06_t_final.log python -m pytest tests -q -m gpu
step 900 gpurun_out/r06_b8.json python bench.py
step 200 gpurun_out/r06_b9.json python bench.py --steps 20 --warmup 5

#!/bin/bash
# round 6, GPU call 8: the whole GPU suite with the speculative lean step as the C host's default
cd "$(dirname "$0")/.." && . tools/gpu_steps.sh
step 1100 gpurun_out/r06_t_all.log python -m pytest tests -q -m gpu

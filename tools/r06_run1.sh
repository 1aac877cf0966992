#!/bin/bash
# round 6, GPU call 1: the density-health statistic (test + bisect over builds), box calibration, phase stamps of the rebuild
cd "$(dirname "$0")/.." && . tools/gpu_steps.sh
L=pi-sph-fluid_amd/csrc
step 300 gpurun_out/r06_health.log python -m pytest tests/test_gpu_health.py -x -q
step 200 gpurun_out/r06_rho_default.json python tools/rho_gate_gpu.py $L/libsph_hip.so 64
step 200 gpurun_out/r06_rho_ieee.json python tools/rho_gate_gpu.py $L/libsph_hip_ieee.so 64
step 200 gpurun_out/r06_rho_skin012.json python tools/rho_gate_gpu.py $L/libsph_hip.so 64 0.12
step 200 gpurun_out/r06_rho_det.json python tools/rho_gate_gpu.py $L/libsph_hip.so 64 default 1
step 200 gpurun_out/r06_b1.json python bench.py --no-cpu --no-also --steps 20 --warmup 5
step 300 gpurun_out/r06_kbench_dev.txt python tools/kbench_gpu.py 4200
step 300 gpurun_out/r06_kbench_200.txt python tools/kbench_gpu.py 300

#!/bin/bash
# round 6, GPU call 22: a sweep of the launch-shape constants on the final library
cd "$(dirname "$0")/.." && . tools/gpu_steps.sh
step 1100 gpurun_out/r06_ab_shapes.txt bash tools/ab_libs.sh "cfg2 5 200;cfg2 1200 1000;cfg2 4000 1000" libsph_hip.so libsph_hip_vw256.so libsph_hip_vw768.so libsph_hip_cw128.so libsph_hip_cw512.so libsph_hip_xc4.so libsph_hip_xc16.so

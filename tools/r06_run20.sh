#!/bin/bash
# round 6, GPU call 20: where in the density launch the verify jobs sit (after 3/8, 4/8, 5/8 of the tiles)
cd "$(dirname "$0")/.." && . tools/gpu_steps.sh
step 900 gpurun_out/r06_ab_verify_at.txt bash tools/ab_libs.sh "cfg2 200 1000;cfg2 1200 1000;cfg2 4000 1000;cfg4 2000 400 --tilt" libsph_hip.so libsph_hip_va4.so libsph_hip_va3.so

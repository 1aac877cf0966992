#!/bin/bash
# round 6, GPU call 4: slab repair scenes, the force pass in reverse tile order, the bound of a step without the gate's launch, the full bench line
cd "$(dirname "$0")/.." && . tools/gpu_steps.sh
step 300 gpurun_out/r06_slab_repair_explore2.txt python tools/slab_repair_explore.py
step 400 gpurun_out/r06_ab_rev.txt bash tools/ab_libs.sh "cfg2 5 200;cfg2 200 1000;cfg2 4000 1000;cfg4 50 200 --tilt" libsph_hip.so libsph_hip_rev.so
step 300 gpurun_out/r06_ab_nogate.txt bash tools/ab_libs.sh "cfg1 200 2000;cfg2 5 200;cfg0 100 4000" libsph_hip.so libsph_hip_nogate.so
step 900 gpurun_out/r06_b3.json python bench.py --no-cpu

#!/bin/bash
# round 6, GPU call 3: where one slab's step goes (kernel trace of the C host with and without graphs), new tests, slab repair scenes
cd "$(dirname "$0")/.." && . tools/gpu_steps.sh
root=$(pwd)
H=$root/pi-sph-fluid_amd/host/slab_sph_fluid
step 600 gpurun_out/r06_t_new.log python -m pytest tests/test_slab_c_host.py -x -q -m gpu -k "graphed or oracle"
step 300 gpurun_out/r06_t_verlet.log python -m pytest tests/test_gpu_verlet.py -x -q -m gpu -k "missing_pairs"
step 300 gpurun_out/r06_slab_repair_explore.txt python tools/slab_repair_explore.py
cd /tmp
for g in 1 0; do
  step 200 $root/gpurun_out/r06_slabtrace_g$g.log rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/slabtrace_g$g -o trace -- $H --ranks 1 --scene dam --steps 300 --warmup 200 --lean-graph $g
  f=$(find /tmp/slabtrace_g$g -name 'trace_kernel_trace.csv' | head -1)
  python3 $root/profiles/gaps.py $f 300 2 k_slab_head > $root/gpurun_out/r06_slab_gaps_g$g.txt 2>&1
  cp $(find /tmp/slabtrace_g$g -name 'trace_kernel_stats.csv' | head -1) $root/gpurun_out/r06_slab_stats_g$g.csv
done
step 200 $root/gpurun_out/r06_steptrace.log rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/steptrace -o trace -- python3 $root/bench.py --no-cpu --no-also --steps 300 --warmup 200
f=$(find /tmp/steptrace -name 'trace_kernel_trace.csv' | head -1)
python3 $root/profiles/gaps.py $f 300 2 k_density_list > $root/gpurun_out/r06_step_gaps.txt 2>&1

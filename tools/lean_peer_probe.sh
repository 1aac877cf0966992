#!/bin/bash
# tools/lean_peer_probe.sh [runs [ranks]] — the lean step over the peer transport with `ranks` processes on ONE GPU (the command of
# tests/test_slab_c_host.py::test_c_host_lean_step_over_peer_mapped_memory_equals_sph_step_bitwise), `runs` times, stopping at the
# first run that fails: its stderr says which wait gave up (FLAG_PEER_DIAG).  (GPU box.)
cd "$(dirname "$0")/.." || exit 1
runs=${1:-6}; ranks=${2:-4}
mkdir -p gpurun_out
for k in $(seq 1 "$runs"); do
    timeout -k 10 120 pi-sph-fluid_amd/host/slab_sph_fluid --ranks "$ranks" --transport peer --lean 1 --one-launch-wgs 256 --block 600 150 90 20 \
        --velocity 5 0 --steps 250 --warmup 50 --deterministic --skin 0 > gpurun_out/lean_probe_$k.out 2> gpurun_out/lean_probe_$k.err
    rc=$?
    echo "run $k: rc=$rc $(grep -o '"timesteps_per_s": [0-9.]*' gpurun_out/lean_probe_$k.out | head -1)"
    if [ $rc -ne 0 ]; then grep -h "sph_sync\|give up\|error" gpurun_out/lean_probe_$k.err | cut -c1-600; exit 1; fi
done
echo "ok: $runs runs"

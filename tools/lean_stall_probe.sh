#!/bin/bash
# tools/lean_stall_probe.sh — the lean step over the peer transport (three ranks on ONE GPU) with rank 1's host held up for 300 us
# between the head kernel and the rest of every third step ($SPH_TEST_STALL_AFTER_HEAD), through the library as built and, if
# present, through `make variant NAME=eqwait VFLAGS="-DSPH_PEER_WAIT_EQUAL -DSPH_TEST_HOOKS"` (the arrival-flag wait of rounds 3-5: gives up).
# Round 6: the stall hook exists in test builds only (-DSPH_TEST_HOOKS): the probe runs host/slab_sph_fluid_stress (`make stress`).  (GPU box.)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out /tmp/eqwait
make -s -C pi-sph-fluid_amd stress
cmd="pi-sph-fluid_amd/host/slab_sph_fluid_stress --lean-graph 0 --lean-spec 1 --ranks 3 --transport peer --lean 1 --one-launch-wgs 256 --block 600 150 90 20 --velocity 5 0 --steps 200 --warmup 40 --deterministic"
for v in asbuilt eqwait; do
    if [ $v = eqwait ]; then
        [ -f pi-sph-fluid_amd/csrc/libsph_hip_eqwait.so ] || continue
        cp pi-sph-fluid_amd/csrc/libsph_hip_eqwait.so /tmp/eqwait/libsph_hip_co.so
        export LD_LIBRARY_PATH=/tmp/eqwait:$LD_LIBRARY_PATH
    fi
    for stall in none 1:300:3; do
        if [ $stall = none ]; then unset SPH_TEST_STALL_AFTER_HEAD; else export SPH_TEST_STALL_AFTER_HEAD=$stall; fi
        timeout -k 10 120 $cmd > gpurun_out/stall_${v}_${stall//:/_}.out 2> gpurun_out/stall_${v}_${stall//:/_}.err
        echo "$v stall=$stall rc=$? $(grep -o '"ticks_per_s": [0-9.]*' gpurun_out/stall_${v}_${stall//:/_}.out | head -1) $(grep -h 'give up' gpurun_out/stall_${v}_${stall//:/_}.err | head -1 | cut -c1-400)"
    done
done

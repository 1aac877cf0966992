#!/bin/bash
# tools/slab_verify_sweep.sh [most ...] — one slab of 2 000 000 particles through the C host (lean step, 8(d) windows: 200 + 5 x 1000):
# without verification, with it, and with it up to `most` queued group pairs ($SPH_SLAB_VERIFY_MOST; more: the rebuild).  (GPU box.)
cd "$(dirname "$0")/.." || exit 1
H=pi-sph-fluid_amd/host/slab_sph_fluid
# $SPH_SWEEP_LIB = a variant build of the library (make variant NAME=x): the host finds it first through LD_LIBRARY_PATH
if [ -n "$SPH_SWEEP_LIB" ]; then mkdir -p /tmp/sweeplib && cp "pi-sph-fluid_amd/csrc/$SPH_SWEEP_LIB" /tmp/sweeplib/libsph_hip.so && export LD_LIBRARY_PATH=/tmp/sweeplib:$LD_LIBRARY_PATH; echo "library: $SPH_SWEEP_LIB"; fi
run() {  # label verify most
    SPH_SLAB_VERIFY_MOST=$3 $H --ranks 1 --scene dam --lean 1 --verify $2 --warmup 200 --steps 1000 --windows 5 2>/dev/null | python3 -c "
import json,sys
d=[json.loads(l) for l in sys.stdin if l.startswith('{')][0]
print('%-22s median %8.1f  windows %s  rebuilds %d' % ('$1', d['ticks_per_s'], ' '.join('%7.0f' % w for w in d['window_ticks_per_s']), d['neighbour_rebuilds']))"
}
for rep in 1 2; do
run "verify 0" 0 0
run "verify 1" 1 0
for m in "$@"; do run "verify 1, most $m" 1 $m; done
done

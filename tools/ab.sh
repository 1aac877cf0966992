#!/usr/bin/env bash
# tools/ab.sh <libA.so> <libB.so> [rounds] — same-box A/B of two builds of libsph_hip.so: bench.py's cfg2 / cfg1 / developed
# figures, alternating A B A B (boxes differ by ~5 %, so only numbers from one gpurun call compare).
a="$1"; b="$2"; rounds="${3:-2}"
for r in $(seq 1 "$rounds"); do
  for lib in "$a" "$b"; do
    python bench.py --no-cpu --lib "$lib" --steps 1000 --warmup 200 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
print('$lib'.split('/')[-1], 'cfg2 %.0f' % d['timesteps_per_s'], 'dens %.1f force %.1f' % (d['kernel_ms']['density_eos']*1e3, d['kernel_ms']['force_kick']*1e3), ' | '.join('%s %.0f' % (a['workload'][:5], a['timesteps_per_s']) for a in d['also']))"
  done
done

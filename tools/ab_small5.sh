# tools/ab_small5.sh lib... : the regimes where a step is short — cfg1 (2000 steps after 200), cfg0 (4000 after 200), cfg2 at rest (the
# driver's 20 steps after 5, and 200 after 5) and cfg2 after 200 — per build, three times, alternating
run() {
python bench.py --no-cpu --no-also --lib pi-sph-fluid_amd/csrc/$1 --workload $2 --warmup $3 --steps $4 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
print('%-22s %-5s w%-5s k%-5s %9.1f steps/s' % ('$1', '$2', '$3', '$4', d['timesteps_per_s']))"
}
for rep in 1 2 3; do
for lib in "$@"; do
run $lib cfg1 200 2000; run $lib cfg0 200 4000; run $lib cfg2 5 20; run $lib cfg2 5 200; run $lib cfg2 200 1000
done; done

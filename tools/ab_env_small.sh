# tools/ab_env_small.sh VAR=VALUE... : cfg1 and the at-rest window of cfg2 with and without the environment settings (same library, same box)
for rep in 1 2; do for v in "" "$@"; do
env $v python bench.py --no-cpu --no-also --workload cfg1 --steps 2000 --warmup 200 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
print('${v:-default}', 'cfg1 %.0f' % d['timesteps_per_s'])"
env $v python bench.py --no-cpu --no-also --steps 200 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
print('${v:-default}', 'cfg2 at rest %.0f' % d['timesteps_per_s'])"
done; done

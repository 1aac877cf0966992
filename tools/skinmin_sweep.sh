for sm in 0.12 0.08 0.05 0.03; do
  echo "skin_min $sm"
  SPH_BENCH_SKIN_MIN=$sm python bench.py --no-cpu --no-also --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print(' rest20', d['timesteps_per_s'], d['kernel_ms']['density_eos']*1e3, d['kernel_ms']['force_kick']*1e3, d['neighbour_rebuilds_per_step'])"
  SPH_BENCH_SKIN_MIN=$sm python bench.py --no-cpu --no-also --steps 1000 --warmup 200 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print(' s200-1200', d['timesteps_per_s'], d['kernel_ms']['density_eos']*1e3, d['kernel_ms']['force_kick']*1e3, d['neighbour_rebuilds_per_step'], d['skin_at_end_fraction_of_2h'])"
done

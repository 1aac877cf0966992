for r in 30 300 1000 300; do
  SPH_BENCH_PRE_REPS=$r python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-also 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['timesteps_per_s'], d['ms_per_step'], d['kernel_ms']['force_kick_at_begin_end'], d['kernel_ms']['density_eos_at_begin_end'])"
done

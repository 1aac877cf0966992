#!/usr/bin/env bash
# profiles/collect.sh <tag> [bench args...] — run on the GPU box (via gpurun): kernel trace + separate PMC passes of bench.py.
# Raw output goes to gpurun_out/prof_<tag>/ (scratch); summarise with profiles/summarize.py and commit the summary.
#   SPH_PROF_WARM / SPH_PROF_STEPS   warm-up / traced steps of the kernel-trace pass (default 100 / 300)
#   SPH_PROF_PMC_WARM                warm-up of the counter passes (default 30; 10 steps are counted)
#   SPH_PROF_PMC_RANGE               counters only for these dispatches of every kernel, e.g. "[4001-4012]" (rocprofv3
#                                    --kernel-iteration-range): a long warm-up under counter collection takes minutes otherwise
#   SPH_PROF_QUICK=1                 skip the HBM-traffic passes
# e.g. developed flow of cfg2:  SPH_PROF_WARM=4000 SPH_PROF_PMC_WARM=4000 SPH_PROF_PMC_RANGE="[4001-4012]" profiles/collect.sh r03_dev
#      cfg4 on one GPU:         SPH_PROF_WARM=50 SPH_PROF_STEPS=100 profiles/collect.sh r03_cfg4 --workload cfg4 --tilt
# The profiled program is python3 itself (bench.py at N = 1 starts no other process): never put a launcher that forks and
# execs (slab_sph_fluid --ranks N without --rank, bench.py --gpus N) under rocprofv3 — profiles/README.md.
set -uo pipefail
tag="${1:-r01}"
shift || true
root="${GRAFT_REPO_ROOT:-$(pwd)}"
out="$root/gpurun_out/prof_$tag"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
export SPH_BENCH_BOX=0      # (bench.py's box calibration — ~100 ms of copy and v_fma kernels — stays out of the traces)
B="python3 $root/bench.py --no-cpu --no-also $*"
W="${SPH_PROF_WARM:-100}"; S="${SPH_PROF_STEPS:-300}"; PW="${SPH_PROF_PMC_WARM:-30}"
R=""; [ -n "${SPH_PROF_PMC_RANGE:-}" ] && R="--kernel-iteration-range ${SPH_PROF_PMC_RANGE}"
# 1. per-kernel time (the summary committed under profiles/)
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o trace -- $B --steps "$S" --warmup "$W" > "$out/trace.log" 2>&1
# 2. instruction mix / stalls (SQ: 8 slots per pass)
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU \
    $R --output-format csv -d "$out/pmc_sq1" -o pmc -- $B --steps 10 --warmup "$PW" > "$out/pmc_sq1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS \
    $R --output-format csv -d "$out/pmc_sq2" -o pmc -- $B --steps 10 --warmup "$PW" > "$out/pmc_sq2.log" 2>&1
if [ "${SPH_PROF_QUICK:-0}" != "1" ]; then
# 3. HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes (TCC has 4 slots: 3 + 2 do not fit together)
rocprofv3 --kernel-trace --pmc FETCH_SIZE $R --output-format csv -d "$out/pmc_fetch" -o pmc -- $B --steps 10 --warmup "$PW" > "$out/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE $R --output-format csv -d "$out/pmc_write" -o pmc -- $B --steps 10 --warmup "$PW" > "$out/pmc_write.log" 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum $R --output-format csv -d "$out/pmc_l2" -o pmc -- $B --steps 10 --warmup "$PW" > "$out/pmc_l2.log" 2>&1
fi
ls -R "$out" | head -50

#!/usr/bin/env bash
# profiles/collect.sh <tag> [bench args...] — run on the GPU box (via gpurun): kernel trace + separate PMC passes of bench.py.
# Raw output goes to gpurun_out/prof_<tag>/ (scratch); summarise with profiles/summarize.py and commit the summary.
set -uo pipefail
tag="${1:-r01}"
shift || true
root="${GRAFT_REPO_ROOT:-$(pwd)}"
out="$root/gpurun_out/prof_$tag"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
B="python3 $root/bench.py --no-cpu --no-also $*"
# 1. per-kernel time (the summary committed under profiles/)
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o trace -- $B --steps 300 --warmup 100 > "$out/trace.log" 2>&1
# 2. instruction mix / stalls (SQ: 8 slots per pass)
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU \
    --output-format csv -d "$out/pmc_sq1" -o pmc -- $B --steps 10 --warmup 30 > "$out/pmc_sq1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS \
    --output-format csv -d "$out/pmc_sq2" -o pmc -- $B --steps 10 --warmup 30 > "$out/pmc_sq2.log" 2>&1
if [ "${SPH_PROF_QUICK:-0}" != "1" ]; then
# 3. HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes (TCC has 4 slots: 3 + 2 do not fit together)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -o pmc -- $B --steps 10 --warmup 30 > "$out/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -o pmc -- $B --steps 10 --warmup 30 > "$out/pmc_write.log" 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$out/pmc_l2" -o pmc -- $B --steps 10 --warmup 30 > "$out/pmc_l2.log" 2>&1
fi
ls -R "$out" | head -50

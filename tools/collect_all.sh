#!/usr/bin/env bash
# tools/collect_all.sh <round tag, e.g. r04> [a|b|c|all] — on the GPU box (a: cfg2 x 3 + cfg1; b: cfg4 x 2 + c; c: one slab through the C host — a
# gpurun call is 20 minutes at most): every rocprofv3 summary the round commits under profiles/
# (kernel trace + separate PMC passes per regime: profiles/collect.sh), developed flows from checkpoints written outside the
# profiler.  Raw output under gpurun_out/prof_<tag>_*; summaries under gpurun_out/<tag>_*_summary.md and gpurun_out/traffic.json.
set -uo pipefail
R="${1:-r04}"
PART="${2:-all}"
root="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$root"
cp profiles/traffic.json gpurun_out/traffic.json
sum() {
    python3 profiles/summarize.py "gpurun_out/prof_$1" "gpurun_out/$1_rocprofv3_summary.md" gpurun_out/traffic.json "$2" "$3" > /dev/null
    cp "gpurun_out/prof_$1/trace/trace_kernel_stats.csv" "gpurun_out/$1_kernel_stats.csv" 2>/dev/null
    rm -rf "gpurun_out/prof_$1"      # (the raw traces are tens of MB each: gpurun merges at most 64 MiB back)
    echo "== $1"; head -14 "gpurun_out/$1_rocprofv3_summary.md"
}
if [ "$PART" != b ]; then
# cfg2: the default window's early part (steps 100-400)
profiles/collect.sh ${R}_cfg2 > /dev/null 2>&1; sum ${R}_cfg2 cfg2 "--steps 300 --warmup 100"
# the driver's command: 20 steps after 5, the fluid at rest
SPH_PROF_WARM=5 SPH_PROF_STEPS=20 SPH_PROF_PMC_WARM=5 profiles/collect.sh ${R}_cfg2_at_rest > /dev/null 2>&1; sum ${R}_cfg2_at_rest cfg2_at_rest "--steps 20 --warmup 5"
# developed flow of cfg2 from a checkpoint after 4000 steps
python3 bench.py --no-cpu --no-also --steps 1 --warmup 4000 --save-state /tmp/ck_cfg2.npz > /dev/null 2>&1
SPH_PROF_WARM=100 SPH_PROF_STEPS=300 SPH_PROF_PMC_WARM=100 profiles/collect.sh ${R}_cfg2_developed --load-state /tmp/ck_cfg2.npz > /dev/null 2>&1; sum ${R}_cfg2_developed cfg2_developed "--load-state <checkpoint after 4000 steps> --steps 300 --warmup 100"
# cfg1
profiles/collect.sh ${R}_cfg1 --workload cfg1 > /dev/null 2>&1; sum ${R}_cfg1 cfg1 "--workload cfg1 --steps 300 --warmup 100"
fi
if [ "$PART" != a ] && [ "$PART" != c ]; then
# cfg4 on one GPU (at rest under the tilt), and developed (checkpoint after 2000 steps)
SPH_PROF_WARM=50 SPH_PROF_STEPS=100 profiles/collect.sh ${R}_cfg4 --workload cfg4 --tilt > /dev/null 2>&1; sum ${R}_cfg4 cfg4 "--workload cfg4 --tilt --steps 100 --warmup 50"
python3 bench.py --no-cpu --no-also --workload cfg4 --tilt --steps 1 --warmup 2000 --save-state /tmp/ck_cfg4.npz > /dev/null 2>&1
SPH_PROF_WARM=30 SPH_PROF_STEPS=100 SPH_PROF_PMC_WARM=30 profiles/collect.sh ${R}_cfg4_developed --workload cfg4 --tilt --load-state /tmp/ck_cfg4.npz > /dev/null 2>&1; sum ${R}_cfg4_developed cfg4_developed "--workload cfg4 --tilt --load-state <checkpoint after 2000 steps> --steps 100 --warmup 30"
fi
if [ "$PART" != a ]; then
# one slab of 2 000 000 particles through the C host, in-process (one rank: the host neither forks nor execs): the lean step's four
# kernels and the three-call step's six, steps 200-1200 of the dam break
cd /tmp && export TMPDIR=/tmp
# (round 6: lean = the fused speculative lean step, the C host's default; lean_spec4 = --lean-spec 1, four launches; lean_plain = --lean-spec 0)
for v in lean lean_spec4 lean_plain three_call; do
    opts="--lean 1"; [ $v = lean_spec4 ] && opts="--lean 1 --lean-spec 1"; [ $v = lean_plain ] && opts="--lean 1 --lean-spec 0"; [ $v = three_call ] && opts="--lean 0"
    mkdir -p "$root/gpurun_out/prof_${R}_slab1_$v"
    rocprofv3 --kernel-trace --stats --output-format csv -d "$root/gpurun_out/prof_${R}_slab1_$v/trace" -o trace -- \
        "$root/pi-sph-fluid_amd/host/slab_sph_fluid" --ranks 1 --scene dam $opts --warmup 200 --steps 1000 > "$root/gpurun_out/prof_${R}_slab1_$v/trace.log" 2>&1
    python3 "$root/profiles/summarize.py" --trace-only "$root/gpurun_out/prof_${R}_slab1_$v" "$root/gpurun_out/${R}_slab1_${v}_rocprofv3_summary.md" \
        "slab_sph_fluid --ranks 1 --scene dam $opts --warmup 200 --steps 1000" > /dev/null
    rm -rf "$root/gpurun_out/prof_${R}_slab1_$v"
    echo "== ${R}_slab1_$v"; head -16 "$root/gpurun_out/${R}_slab1_${v}_rocprofv3_summary.md"
done
cd "$root"
fi
ls -la gpurun_out/*_summary.md

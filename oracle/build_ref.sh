#!/usr/bin/env bash
# build_ref.sh — TEST INFRASTRUCTURE ONLY.
#
# Compiles the hot-path section of the real reference, straight from where it
# lies under /root/reference, into oracle/_ref/ (git-ignored; travels to the GPU
# box as a prebuilt .so).  No reference source is copied into the repo and no
# stand-in header/library is written:
#
#   * The reference translation unit as a whole is UNBUILDABLE here: line 8
#     includes <ssd1306.h> from an un-vendored, empty git submodule
#     (/root/reference/.gitmodules:1-3) and the display thread (:466-470) calls
#     into it.  We do not stub it.
#   * The hot path (constants/types/kernel maths/neighbour
#     search/SPH sums/physics passes/metaballs = pi_sph_fluid.c:10-411) has no
#     dependency on that library.  Those lines are piped (sed -n) into gcc
#     together with oracle/ref_harness.c (our driver).  The three pieces of
#     main() that define the scene (:476-540), one time step (:612-644) and the
#     pixel grid (:571-577) are spliced, verbatim and at build time only, into
#     wrapper functions whose signatures supply the locals main() declares.
#
# Two builds, as SURVEY.md §7 step 0:
#   libpisph_ref_strict.so : -O2, no -march, no fast-math  (bit-reproducible IEEE f32)
#   libpisph_ref_fast.so   : -Ofast -march=native (the reference's own Makefile:2-4 flags)
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
ref="${PISPH_REFERENCE:-/root/reference}/pi_sph_fluid.c"
out="$here/_ref"
if [ ! -f "$ref" ]; then
    echo "build_ref.sh: $ref not present (GPU box?) — keeping prebuilt oracle/_ref as is" >&2
    exit 0
fi
mkdir -p "$out"

# sanity: the line anchors below must still be what SURVEY.md cites
sed -n '8p'   "$ref" | grep -q 'ssd1306.h'            || { echo "anchor :8 moved"   >&2; exit 1; }
sed -n '10p'  "$ref" | grep -q '#define REALTIME'     || { echo "anchor :10 moved"  >&2; exit 1; }
sed -n '411p' "$ref" | grep -q '^}'                   || { echo "anchor :411 moved" >&2; exit 1; }
sed -n '476p' "$ref" | grep -q 'int particle_counter' || { echo "anchor :476 moved" >&2; exit 1; }
sed -n '612p' "$ref" | grep -q 'pragma omp single'    || { echo "anchor :612 moved" >&2; exit 1; }
sed -n '643p' "$ref" | grep -q 'clock_gettime'        || { echo "anchor :643 moved" >&2; exit 1; }

gen() {
    # standard headers of :1-7 minus <ssd1306.h> (:8); unistd is not needed on the path
    printf '#include <stdio.h>\n#include <stdlib.h>\n#include <math.h>\n#include <pthread.h>\n#include <time.h>\n#include <limits.h>\n'
    printf '#line 10 "%s"\n' "$ref"
    sed -n '10,411p' "$ref"
    printf '#line 1 "%s/ref_harness.c"\n' "$here"
    cat "$here/ref_harness.c"
    # --- scene: main() :476-540 ---
    printf 'int ref_scene_body(struct particle **fluid_out, int *n_fluid_out, struct particle **boundary_out, int *n_boundary_out){\n'
    printf '#line 476 "%s"\n' "$ref"
    sed -n '476,540p' "$ref"
    printf '*fluid_out = fluid; *n_fluid_out = n_fluid; *boundary_out = boundary; *n_boundary_out = n_boundary; free(du_dt); free(dv_dt); return 0; }\n'
    # --- one time step: main() :612-644 (two omp single blocks + three physics passes) ---
    printf 'void ref_step_body(int n_fluid, struct particle *fluid, float *du_dt, float *dv_dt, struct particle *boundary, struct neighbors_context *ctx_fluid, struct neighbors_context *ctx_boundary, float2 g){ struct timespec now;\n'
    printf '#line 612 "%s"\n' "$ref"
    sed -n '612,644p' "$ref"
    printf '(void)now; }\n'
    # --- pixel pseudo-particles: main() :571-577 ---
    printf 'void ref_pixels_body(struct particle *pixel_pseudoparticles){\n'
    printf '#line 571 "%s"\n' "$ref"
    sed -n '571,577p' "$ref"
    printf '}\n'
}

common="-x c - -shared -fPIC -fopenmp -lm -pthread -w"
gen | gcc -O2 $common -o "$out/libpisph_ref_strict.so"
gen | gcc -Ofast -march=native $common -o "$out/libpisph_ref_fast.so"
echo "built $out/libpisph_ref_strict.so $out/libpisph_ref_fast.so"

#!/usr/bin/env bash
# Build the REAL reference (project/sequential/FluidSequential.c) as shared
# libraries, one per (N, Jacobi sweeps), straight from where the source lies
# under /root/reference.  TEST INFRASTRUCTURE: used to pin oracle/fluid_oracle.c
# and to generate tests/golden/; optionally timed as bench.py's cpu_baseline.
#
# N and the sweep count are unguarded literals in the reference
# (FluidSequential.c:6 "#define N 8190", :91 "k < 40"), so -D cannot override
# them: the source is stream-edited by sed on its way into gcc.  Nothing from
# the reference is written into the repo except the compiled objects under
# oracle/_ref/ (git-ignored, but shipped to the GPU box like our own .so).
#
# usage: oracle/build_ref.sh            # default set
#        oracle/build_ref.sh 30:40 ...  # explicit N:iters pairs
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
src="${FLUID_REFERENCE_ROOT:-/root/reference}/project/sequential/FluidSequential.c"
out="$here/_ref"
if [ ! -f "$src" ]; then
    echo "build_ref: $src not present (GPU box?) - keeping prebuilt files" >&2
    exit 0
fi
mkdir -p "$out"
pairs=("$@")
if [ ${#pairs[@]} -eq 0 ]; then
    pairs=(14:40 30:40 61:40 126:40 126:20 254:40 1022:40 4094:40 8190:40 16382:40)
fi
for pr in "${pairs[@]}"; do
    n="${pr%%:*}"; k="${pr##*:}"
    so="$out/libfluidref_n${n}_k${k}.so"
    # canonical oracle flags: -O2, no -march/-mfma, contraction off
    sed -e "s/^#define N 8190 .*/#define N ${n}/" -e "s/k < 40/k < ${k}/" "$src" |
        gcc -x c -O2 -ffp-contract=off -fPIC -shared -w -Dmain=fluidref_main \
            -o "$so" -
    echo "built $so"
done

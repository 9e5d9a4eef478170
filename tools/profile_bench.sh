#!/bin/bash
# rocprofv3 passes over bench.py on the GPU box (run through gpurun), one counter group per pass as the MI355X guide
# prescribes: kernel trace + stats, FETCH_SIZE, WRITE_SIZE, and the SQ / GRBM group (vector instructions, clock).
#   tools/profile_bench.sh GRID [extra bench.py flags]     -> gpurun_out/prof_<GRID>[_<flags>]/{trace,fetch,write,sq}
# then, back in the dev container:  tools/summarize_profiles.py r02 gpurun_out/prof_<GRID>   (-> profiles/)
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
GRID=$1; shift
TAG=$(echo "$GRID $*" | tr -s ' -' '_' | sed 's/_$//')
OUT=$R/gpurun_out/prof_$TAG
ARGS="--grid $GRID --steps 6 --warmup 3 --no-cpu-baseline --no-scaling-base $*"
case " $* " in *only-ordinary*) ;; *) ARGS="$ARGS --no-ordinary";; esac
mkdir -p $OUT
# the hash of the kernel sources this profile is taken from (bench.py checks it against the library it runs with)
(cd $R && python3 -c "import bench; print(bench.library_source_hash())") > $OUT/source_sha256
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o p --output-format csv -- python3 $R/bench.py $ARGS > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o p --output-format csv -- python3 $R/bench.py $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o p --output-format csv -- python3 $R/bench.py $ARGS > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES \
    -d $OUT/sq -o p --output-format csv -- python3 $R/bench.py $ARGS > $OUT/sq.log 2>&1
grep -h '^{' $OUT/trace.log | tail -1 > $OUT/bench_line.json || true
# keep what travels back small: the per-dispatch traces of the counter passes are not needed
rm -f $OUT/fetch/p_kernel_trace.csv $OUT/write/p_kernel_trace.csv
echo "profiles in $OUT"

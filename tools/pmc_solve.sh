# rocprofv3 kernel traces of tools/prof_solve.py configurations -> gpurun_out/kt_<config>/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "$@"; do
  tag=$(echo $cfg | tr ' ' '_')
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_$tag -o p --output-format csv -- python3 $R/tools/prof_solve.py $cfg >> $R/gpurun_out/kt_runs.log 2>&1 || exit 1
done

cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU -d $R/gpurun_out/sq8192 -o p --output-format csv -- python3 $R/bench.py --grid 8192 --steps 4 --warmup 2 --no-cpu-baseline --no-scaling-base --no-ordinary > $R/gpurun_out/sq8192.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAVES TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr -d $R/gpurun_out/sq8192b -o p --output-format csv -- python3 $R/bench.py --grid 8192 --steps 4 --warmup 2 --no-cpu-baseline --no-scaling-base --no-ordinary >> $R/gpurun_out/sq8192.log 2>&1

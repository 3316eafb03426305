cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="python bench.py --steps 1 --warmup 0 --no-cpu-baseline"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAIT_ANY -d gpurun_out/pmc_$1/sq -o sq --output-format csv -- $B > /dev/null 2>&1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum -d gpurun_out/pmc_$1/l2 -o l2 --output-format csv -- $B > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_$1/fetch -o fetch --output-format csv -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_$1/write -o write --output-format csv -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TA_BUSY_avr GRBM_GUI_ACTIVE -d gpurun_out/pmc_$1/sq2 -o sq2 --output-format csv -- $B > /dev/null 2>&1
ls -R gpurun_out/pmc_$1 | head -40

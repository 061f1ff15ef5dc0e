#!/bin/bash
# fr/fft under rocprofv3 on the GPU box: kernel trace + separate PMC passes (SQ issue / wait buckets, SQ instruction mix,
# LDS conflicts, FETCH_SIZE, WRITE_SIZE) of ONE command (tools/bench_fft.py bn254 20 24) -> gpurun_out/$1/ ;
# tools/fft_stats_md.py turns the CSVs into profiles/rNN_fft_stats.md.  PMC passes never carry a trace flag (gpurun rule).
out=/root/repo/gpurun_out/${1:-fft_prof}
shift
CMD="python /root/repo/tools/bench_fft.py ${*:-bn254 20 24}"
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/trace -o t --output-format csv -- $CMD > $out/trace.log 2>&1
pass() {  # name, counters...
  name=$1; shift
  rocprofv3 --pmc "$@" -d $out/$name -o p --output-format csv -- $CMD > $out/$name.log 2>&1
}
pass sq_time SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
pass sq_insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE
pass fetch FETCH_SIZE
pass write WRITE_SIZE
find $out -name "*agent_info.csv" -delete
find $out -name "*kernel_trace.csv" -size +8M -delete
du -sh $out

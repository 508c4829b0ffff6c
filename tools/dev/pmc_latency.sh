# Development helper (GPU box): where the cycles of the latency-bound kernels go (small-MSM tree, sixteen-lane Miller lines): one PMC counter
# per run over tools/dev/small_msm_prof.py and tools/dev/ml_loop.py, summarised per kernel in gpurun_out/<TAG>_pmc_latency.txt.
TAG=${TAG:-r04}
O=/root/repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
for C in SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_HITS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE; do
  timeout 200 rocprofv3 --pmc $C --output-format csv -d $O/${TAG}_pl_sm_$C -- python /root/repo/tools/dev/small_msm_prof.py > /dev/null 2>&1
  timeout 200 rocprofv3 --pmc $C --output-format csv -d $O/${TAG}_pl_ml_$C -- python /root/repo/tools/dev/ml_loop.py > /dev/null 2>&1
done
python /root/repo/tools/pmc_summary.py $O/${TAG}_pl_sm_* $O/${TAG}_pl_ml_* > $O/${TAG}_pmc_latency.txt
rm -rf $O/${TAG}_pl_sm_* $O/${TAG}_pl_ml_*

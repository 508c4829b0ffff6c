# Development helper (GPU box): kernel statistics and counters of the witness map (the NTT passes k_ntt_r4, the sparse mat-vec) at D = 2^20,
# one counter per pass as MI355X_MICROARCH.md prescribes
O=/root/repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r05_prof_wm -- python /root/repo/tests/perf/qap_perf.py > $O/r05_witness_map_under_rocprof.txt 2>/dev/null
cp $O/r05_prof_wm/*/*kernel_stats.csv $O/r05_kernel_stats_witness_map.csv
for C in SQ_INSTS_VALU GRBM_GUI_ACTIVE FETCH_SIZE WRITE_SIZE SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR; do
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $O/r05_pmc_wm_$C -- python /root/repo/tests/perf/qap_perf.py > /dev/null 2>&1
done
python /root/repo/tools/pmc_summary.py $O/r05_pmc_wm_* | grep "ntt\|csr\|fr_" > $O/r05_pmc_witness_map.txt
rm -rf $O/r05_prof_wm $O/r05_pmc_wm_*
cat $O/r05_witness_map_under_rocprof.txt; cut -c1-60,150-260 $O/r05_kernel_stats_witness_map.csv | head -8; cat $O/r05_pmc_witness_map.txt

# Development helper (GPU box): kernel trace of the prover's loop (tests/perf/prove_perf.py) -> gpurun_out/<TAG>_kernel_trace_prove.csv and the timings
TAG=${TAG:-prove}
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /root/repo/gpurun_out/prof_prove -- python /root/repo/tests/perf/prove_perf.py > /root/repo/gpurun_out/${TAG}_prove_perf.txt 2>&1
cp /root/repo/gpurun_out/prof_prove/*/*kernel_trace.csv /root/repo/gpurun_out/${TAG}_kernel_trace_prove.csv; rm -rf /root/repo/gpurun_out/prof_prove
grep median /root/repo/gpurun_out/${TAG}_prove_perf.txt | tail -2

# Development helper (GPU box): the MSM parity tests, then the headline loop A/B on ONE box (boxes differ by +-4 %): with DGPU_LIB_OLD=<an older build of the
# library> the default bench line and the one-in-flight line alternate between the two libraries (results under gpurun_out/ab_*).
set -x
export TMPDIR=/tmp
O=/root/repo/gpurun_out
timeout 1200 python -m pytest tests/test_gpu_msm.py tests/test_gpu_full_sizes.py tests/test_gpu_precomputed.py -x -q -m gpu 2>&1 | tail -3 > $O/ab_tests.txt
rm -f $O/ab_bench_new.json $O/ab_bench_old.json
for rep in 1 2 3; do
  python bench.py --steps 20 --no-secondary --no-cpu-baseline >> $O/ab_bench_new.json 2>/dev/null
  if [ -n "$DGPU_LIB_OLD" ]; then DGPU_LIB=$DGPU_LIB_OLD python bench.py --steps 20 --no-secondary --no-cpu-baseline >> $O/ab_bench_old.json 2>/dev/null; fi
done
K=6 python tools/dev/g2_loop.py > $O/ab_g2_new.txt 2>&1
if [ -n "$DGPU_LIB_OLD" ]; then DGPU_LIB=$DGPU_LIB_OLD K=6 python tools/dev/g2_loop.py > $O/ab_g2_old.txt 2>&1; fi

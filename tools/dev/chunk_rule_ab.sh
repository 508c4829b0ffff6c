# Development helper (GPU box): the run-length term of the chunk rule (choose_chunk / k_dyn_chunk) against a library without it (DGPU_LIB_OLD):
# MSM parity tests, the witness-shaped and dense MSMs on width-17 tables, the default bench line (headline + prover), interleaved on one box
cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_msm.py tests/test_gpu_full_sizes.py tests/test_gpu_precomputed.py tests/test_gpu_prove_abi.py -x -q -m gpu 2>&1 | tail -1
for rep in 1 2; do
  for lib in new old; do
    if [ $lib = old ]; then export DGPU_LIB=$DGPU_LIB_OLD; else unset DGPU_LIB; fi
    echo "== $lib"
    CS=17 python tests/perf/witness_msm_perf.py 2>&1 | grep "c=17"
    python bench.py --no-cpu-baseline --no-cpu-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['secondary']
print('bench', d['value'], d['latency_ms_one_in_flight'], s['g2_msm_ms'], s['prove_2p20_ms'], s['prove_2p20_ms_per_proof_4_in_flight'], s['g1_2p24_single_gpu']['latency_ms'])"
  done
done

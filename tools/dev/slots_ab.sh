# Development helper (GPU box): a library built with more slots per context (calls in flight) against DGPU_LIB_OLD (six): the headline loop with
# 6 / 8 / 12 calls in flight and the default bench line's prover figures
cd /root/repo
for rep in 1 2; do
  for lib in new old; do
    if [ $lib = old ]; then export DGPU_LIB=$DGPU_LIB_OLD; else unset DGPU_LIB; fi
    for inf in 6 8 12; do
      python bench.py --steps 24 --no-secondary --no-cpu-baseline --inflight $inf 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib inflight=$inf', d['value'], d['ms_per_step'])"
    done
    python bench.py --no-cpu-baseline --no-cpu-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['secondary']
print('$lib bench', d['value'], s['g2_msm_ms_per_msm_4_in_flight'], s['miller_loop_1024_pairs_ms_per_call_6_in_flight'], s['prove_2p20_ms'], s['prove_2p20_ms_per_proof_4_in_flight'], s['snarkpack_aggregate_1024_proofs_ms'])"
  done
done

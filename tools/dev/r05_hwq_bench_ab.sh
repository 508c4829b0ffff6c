# Development helper (GPU box): the whole bench line under ROCm's default 4 hardware queues and under GPU_MAX_HW_QUEUES=8, interleaved on one box
cd /root/repo
for rep in 1 2; do for q in 4 8; do
  GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu-baseline --no-cpu-legs 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); s=d['secondary']
ks=['g2_msm_ms','miller_loop_1024_pairs_ms','miller_loop_1024_pairs_ms_per_call_6_in_flight','verify_one_proof_ms','verify_1024_proofs_one_call_ms','snarkpack_aggregate_1024_proofs_native_transcript_ms','snarkpack_verify_aggregate_native_transcript_ms','witness_map_ms','prove_2p20_ms','prove_2p20_ms_per_proof_4_in_flight']
print('hwq=$q', d['value'], d['ms_per_step'], d['latency_ms_one_in_flight'], ' '.join('%s=%s' % (k.replace('_ms','').replace('snarkpack_','').replace('_native_transcript',''), s[k]) for k in ks))"
done; done

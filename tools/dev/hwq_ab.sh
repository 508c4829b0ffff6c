# Development helper (GPU box): the headline loop under different numbers of hardware queues (ROCm's GPU_MAX_HW_QUEUES, default 4) and calls in flight
cd /root/repo
for rep in 1 2; do
for q in 4 8; do
  for inf in 4 6 8; do
    GPU_MAX_HW_QUEUES=$q python bench.py --steps 20 --no-secondary --no-cpu-baseline --inflight $inf 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('hwq=$q inflight=$inf', d['value'], d['ms_per_step'])
"
  done
done
done

cd /root/repo
for rep in 1 2; do
for rs in -1 4 5; do
  for lib in new old; do
    if [ $lib = old ]; then export DGPU_LIB=/root/repo/crypto_amd/libdock_gpu_old.so; else unset DGPU_LIB; fi
    python bench.py --steps 20 --no-secondary --no-cpu-baseline --reduce-shift $rs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib rs=$rs', d['value'], d['ms_per_step'], d.get('stages_ms_one_in_flight'), d.get('latency_ms'))
"
  done
done
done

# Development helper (GPU box): k_accumulate duration with one call in flight (stage timers of the twin), a few runs; then the default line twice
cd /root/repo
for i in 1 2 3; do python bench.py --no-secondary --no-cpu-baseline --inflight 1 --reduce-lanes 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('inflight1', d['value'], d['stages_ms_one_in_flight'])"; done
for i in 1 2; do python bench.py --no-secondary --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default', d['value'], d['ms_per_step'], d['latency_ms_one_in_flight'])"; done
K=6 python tools/dev/g2_loop.py

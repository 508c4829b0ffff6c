# Development helper (GPU box): same-box A/B of the default bench line (product library) against the twin with a knob set; ARGS="..." variants
cd /root/repo
for v in "" "--reduce-lanes 4" "--reduce-lanes 0" "" "--reduce-lanes 4" "--reduce-lanes 0"; do python bench.py --no-secondary --no-cpu-baseline $v 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$v]', d['value'], d['ms_per_step'], d['latency_ms_one_in_flight'], d.get('library','product'))"; done

cd /root/repo
python -m pytest tests/test_gpu_precomputed.py -x -q -m gpu 2>&1 | tail -5
for l in 4 2 0 4 2 0; do python bench.py --no-secondary --no-cpu-baseline --reduce-lanes $l 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lanes',$l, d['value'], d['ms_per_step'], d.get('stages_ms_one_in_flight'))"; done

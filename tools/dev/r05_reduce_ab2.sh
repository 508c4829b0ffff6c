# Development helper (GPU box): the bit-marginal reduction with 8 / 16 buckets per lane under six calls in flight; G2 scan form against marginals
cd /root/repo
for sh in -1 3 4 -1 3 4; do python bench.py --no-secondary --no-cpu-baseline --reduce-shift $sh 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('shift',$sh, d['value'], d['ms_per_step'])"; done
for l in 4 0 4 0; do echo "G2 lanes $l"; K=8 REDUCE_LANES=$l python tools/dev/g2_loop.py; done

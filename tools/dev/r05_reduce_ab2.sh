# Development helper (GPU box): the bit-marginal reduction with 8 / 16 buckets per lane under six calls in flight; G2 scan form against marginals
cd /root/repo
for sh in -1 3 4 -1 3 4; do python bench.py --no-secondary --no-cpu-baseline --reduce-shift $sh 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('shift',$sh, d['value'], d['ms_per_step'])"; done
python - <<'PY'
import os, sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools/dev')
os.environ['K'] = '8'
from crypto_amd._native import lib
import runpy
for l in (4, 0, 4, 0):
    os.environ['REDUCE_LANES'] = str(l)
    import crypto_amd as ca
    ca.init(0); lib().dgpu_set_reduce_lanes(l)
    print('G2 lanes', l); runpy.run_path('/root/repo/tools/dev/g2_loop.py')
PY

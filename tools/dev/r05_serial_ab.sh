# Development helper (GPU box): the field products with carry-seeded column chains (-DFS_SERIAL_LOW -DFS_ALT_HIGH) against the default build.
# 1. tools/ubench/bin/madd_* (built on the dev box from tools/ubench/madd_rate.hip)   2. bench.py stage timers / default line with the alternative libraries
cd /root/repo
# the alternative build (out of tree: the default objects stay as they are), unless the dev box already put it there
if [ ! -f crypto_amd/alt_ser2/libdock_gpu.so ]; then
  rm -rf /tmp/build_ser2 && mkdir -p /tmp/build_ser2/crypto_amd /tmp/build_ser2/include crypto_amd/alt_ser2 && cp -r crypto_amd/csrc /tmp/build_ser2/crypto_amd/ && cp include/*.h include/*.hpp /tmp/build_ser2/include/
  (cd /tmp/build_ser2/crypto_amd/csrc && rm -f *.o && make -j16 FLAGS="-O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wno-pass-failed -Xarch_host -mbmi2 -Xarch_host -madx -DFS_SERIAL_LOW -DFS_ALT_HIGH" > /tmp/build_ser2/log.txt 2>&1) && cp /tmp/build_ser2/crypto_amd/*.so crypto_amd/alt_ser2/
fi
mkdir -p tools/ubench/bin
for v in "base:" "alt:-DFS_ALT_HIGH" "ser:-DFS_SERIAL_LOW" "ser2:-DFS_SERIAL_LOW -DFS_ALT_HIGH"; do n=${v%%:*}; [ -x tools/ubench/bin/madd_$n ] || hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-pass-failed ${v#*:} tools/ubench/madd_rate.hip -o tools/ubench/bin/madd_$n 2>/dev/null; done
for v in base alt ser ser2; do echo "== ubench $v"; tools/ubench/bin/madd_$v | grep "madd_s"; done
run() { # $1 label, env already set
  for i in 1 2; do python bench.py --no-secondary --no-cpu-baseline --inflight 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 inflight1', d['value'], d['stages_ms_one_in_flight'])"; done
  for i in 1 2; do python bench.py --no-secondary --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 default', d['value'], d['ms_per_step'], d['latency_ms_one_in_flight'], d['config']['bit_exact_vs_closed_form'])"; done
}
run base
export DGPU_LIB=/root/repo/crypto_amd/alt_ser2/libdock_gpu.so DGPU_DEV_LIB=/root/repo/crypto_amd/alt_ser2/libdock_gpu_dev.so
run ser2
python -m pytest tests/test_gpu_msm.py tests/test_gpu_precomputed.py tests/test_gpu_pairing.py -x -q -m gpu 2>&1 | tail -3
unset DGPU_LIB DGPU_DEV_LIB
run base

DGPU_NTT_OCC=1 python tests/perf/qap_perf.py 2>&1 | grep -m1 occupancy
for g in 128 256 384 512 768 1024 1536; do echo "== tile=11 grid=$g"; DGPU_NTT_GRID=$g python tests/perf/qap_perf.py; done
for g in 256 512 768 1024 2048 3072; do echo "== tile=10 grid=$g"; DGPU_NTT_TILE_LOG=10 DGPU_NTT_GRID=$g python tests/perf/qap_perf.py; done

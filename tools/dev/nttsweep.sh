python -m pytest tests/test_gpu_witness_map.py tests/test_gpu_full_sizes.py -x -q -m gpu 2>&1 | tail -2
DGPU_NTT_TILE_LOG=11 python -m pytest tests/test_gpu_witness_map.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2 3; do python tests/perf/qap_perf.py; done
DGPU_NTT_TILE_LOG=11 python tests/perf/qap_perf.py
for l in 10 11 12 14 16 18 21 22; do LOG2N=$l python tests/perf/qap_perf.py; done

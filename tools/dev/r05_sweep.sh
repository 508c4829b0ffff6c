# Development helper (GPU box): with the stable harness, calls in flight x reduction shape (twin: the knob), two passes
cd /root/repo
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["ms_per_step"])'
for pass in 1 2; do
for k in 4 6 8; do for sh in -1 3 4; do
  a="--inflight $k"; [ $sh -ge 0 ] && a="$a --reduce-shift $sh" || a="$a --reduce-lanes 0"
  python bench.py --no-secondary --no-cpu-baseline $a 2>/dev/null | tail -1 | python -c "$P" "inflight=$k shift=$sh"
done; done; done

"""cProfile of aggregate_proofs at N proofs (development helper)."""
import os, sys, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests", "perf"))
os.environ.setdefault("N", "1024")
src = open(os.path.join(ROOT, "tests", "perf", "bench_aggregation.py")).read().split("ca.prof.enable(True)")[0]
g = {"__name__": "x", "__file__": os.path.join(ROOT, "tests", "perf", "bench_aggregation.py")}
exec(compile(src, "bench_setup", "exec"), g)
AG, pk, proofs = g["AG"], g["pk"], g["proofs"]
AG.aggregate_proofs(pk, AG.MerlinTranscript(b"bench"), proofs)
pr = cProfile.Profile(); pr.enable()
AG.aggregate_proofs(pk, AG.MerlinTranscript(b"bench"), proofs)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)

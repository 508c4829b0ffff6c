import os, sys, time, cProfile, pstats
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle"); sys.path.insert(0, "/root/repo/tests/perf")
os.environ["N"] = "1024"
import runpy
# reuse the bench script's setup by exec'ing it up to the aggregate
src = open("/root/repo/tests/perf/bench_aggregation.py").read().split("ca.prof.enable(True)")[0]
g = {"__name__": "x", "__file__": "/root/repo/tests/perf/bench_aggregation.py"}
exec(compile(src, "bench_setup", "exec"), g)
AG, pk, vsrs, proofs, vk, inputs, rnd = g["AG"], g["pk"], g["vsrs"], g["proofs"], g["vk"], g["inputs"], g["rnd"]
agg = AG.aggregate_proofs(pk, AG.MerlinTranscript(b"bench"), proofs)
AG.verify_aggregate_proof(vsrs, {"vk": vk}, inputs, agg, rnd(), AG.MerlinTranscript(b"bench"))
pr = cProfile.Profile(); pr.enable()
AG.verify_aggregate_proof(vsrs, {"vk": vk}, inputs, agg, rnd(), AG.MerlinTranscript(b"bench"))
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)

import sys; sys.path[:0]=["/root/repo","/root/repo/oracle","/root/repo/tests"]
import time, threading, numpy as np, oracle_c as O, crypto_amd as ca, bench as B
from crypto_amd import pairing, fixed_base as FB
sys.setswitchinterval(1e-4)
ca.init(0); n=1024
with FB.WindowTable(ca.G1, O.G1.generator()) as t1, FB.WindowTable(ca.G2, O.G2.generator()) as t2:
    A,_=t1.multiply_many(B.seeded_scalars(1,n)); Q,_=t2.multiply_many(B.seeded_scalars(2,n)); Cc,_=t1.multiply_many(B.seeded_scalars(4,n))
m=B.seeded_scalars(3,n)
pc=pairing.G2Prepared.from_affine(Q[:2])
def timed(f,k=30):
    for _ in range(8): f()
    t0=time.perf_counter()
    for _ in range(k): f()
    return (time.perf_counter()-t0)/k*1e3
def branch_a(): return pairing.multi_miller_loop_scaled(A,m,Q)
def branch_b():
    r=[None,None]
    def w(i): r[i]=ca.msm_bigint(ca.G1,Cc,m)
    ts=[threading.Thread(target=w,args=(i,)) for i in range(2)]
    [t.start() for t in ts]; [t.join() for t in ts]
    return pairing.multi_miller_loop(A[:2], pc)
def both():
    ts=[threading.Thread(target=branch_a), threading.Thread(target=branch_b)]
    [t.start() for t in ts]; [t.join() for t in ts]
print("branch a (scaled Miller) alone: %.3f" % timed(branch_a))
print("branch b (2 MSMs, then 2 prepared pairs) alone: %.3f" % timed(branch_b))
print("msm 1024 alone: %.3f" % timed(lambda: ca.msm_bigint(ca.G1,Cc,m)))
print("2 prepared pairs alone: %.3f" % timed(lambda: pairing.multi_miller_loop(A[:2], pc)))
print("both in parallel: %.3f" % timed(both))
def two_msms():
    ts=[threading.Thread(target=lambda: ca.msm_bigint(ca.G1,Cc,m)) for _ in range(2)]
    [t.start() for t in ts]; [t.join() for t in ts]
def two_noop():
    ts=[threading.Thread(target=lambda: None) for _ in range(2)]
    [t.start() for t in ts]; [t.join() for t in ts]
print("two MSMs on two threads: %.3f" % timed(two_msms))
print("two empty threads: %.3f" % timed(two_noop))
print("two MSMs one after the other: %.3f" % timed(lambda: (ca.msm_bigint(ca.G1,Cc,m), ca.msm_bigint(ca.G1,Cc,m))))
acc=[0.0,0.0]
for _ in range(30):
    t0=time.perf_counter(); two_msms(); t1=time.perf_counter(); pairing.multi_miller_loop(A[:2], pc); t2=time.perf_counter()
    acc[0]+=t1-t0; acc[1]+=t2-t1
print("branch b split: MSM pair %.3f, prepared pairs after it %.3f" % (acc[0]/30*1e3, acc[1]/30*1e3))
acc=[0.0,0.0]
for _ in range(30):
    t0=time.perf_counter(); ca.msm_bigint(ca.G1,Cc,m); t1=time.perf_counter(); pairing.multi_miller_loop(A[:2], pc); t2=time.perf_counter()
    acc[0]+=t1-t0; acc[1]+=t2-t1
print("one MSM on this thread %.3f, prepared pairs after it %.3f" % (acc[0]/30*1e3, acc[1]/30*1e3))

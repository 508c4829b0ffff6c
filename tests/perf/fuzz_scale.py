"""One-off randomized cross-check of the batched scaling (GLV), folding (mul_add, G1 and G2) and fixed-base kernels against the oracle."""
import sys
sys.path.insert(0,"/root/repo"); sys.path.insert(0,"/root/repo/oracle")
import numpy as np, crypto_amd as ca, oracle_c as O
from crypto_amd import pairing_check as pc, fixed_base as fb
from crypto_amd.aggregation import ops
ca.init(0)
R=ops.R_MOD
k0=O.rand_scalars(1,1)[0]; d=O.rand_scalars(2,1)[0]
bad=0
for seed in range(3):
    n=400
    P1=O.G1.gen_seq(k0,d,n,threads=16); P2=O.G2.gen_seq(d,k0,n,threads=16)
    sc=O.rand_scalars(50+seed,n); sv=[O.limbs_to_int(s) for s in sc]
    # small / sparse scalars mixed in
    for i in range(0,n,7): sv[i]=sv[i]>>(i%250)
    # scalars built from base-|x| digits (the G2 split) and from (k1, k2) around the GLV lattice borders: zero / one / maximal digits
    X=0xD201000000010000; LAM=X*X-1; rng=np.random.default_rng(seed)
    for i in range(3,n,5):
        dg=[int(rng.choice([0,1,X-1,int(rng.integers(0,2**63))])) for _ in range(4)]; dg[3]=dg[3]%(R//X**3)
        sv[i]=(dg[0]+dg[1]*X+dg[2]*X*X+dg[3]*X**3)%R
    for i in range(4,n,11):
        sv[i]=(int(rng.choice([0,1,2**127,2**128-1]))+int(rng.choice([0,1,2**126,LAM-1]))*LAM)%R
    out,inf=pc.g1_scale_each(P1,np.stack([O.int_to_limbs(v,4) for v in sv]))
    m1=ops.mul_add(ca.G1,P1,sv,P1[::-1].copy()); m2=ops.mul_add(ca.G2,P2,sv,P2[::-1].copy())
    with fb.WindowTable(ca.G2,P2[0]) as t: f2,_=t.multiply_many(sv)
    for i in range(n):
        e=O.G1.to_affine(O.G1.mul(P1[i],O.int_to_limbs(sv[i],4)))
        if not (bool(inf[i])==e[1] and (e[1] or (out[i]==e[0]).all())): bad+=1; print("scale",seed,i)
        e=O.G1.to_affine(O.G1.add(O.G1.mul(P1[i],O.int_to_limbs(sv[i],4)),O.G1.mul(P1[n-1-i],O.int_to_limbs(1,4))))
        if not (e[1] and not m1[i].any() or (m1[i]==e[0]).all()): bad+=1; print("muladd1",seed,i)
        e=O.G2.to_affine(O.G2.add(O.G2.mul(P2[i],O.int_to_limbs(sv[i],4)),O.G2.mul(P2[n-1-i],O.int_to_limbs(1,4))))
        if not (e[1] and not m2[i].any() or (m2[i]==e[0]).all()): bad+=1; print("muladd2",seed,i)
        e=O.G2.to_affine(O.G2.mul(P2[0],O.int_to_limbs(sv[i],4)))
        if not (e[1] and not f2[i].any() or (f2[i]==e[0]).all()): bad+=1; print("fixed2",seed,i)
# one scalar for all points (scalar_stride 0: the folding step of the aggregation)
for seed in range(20):
    n=48; rng=np.random.default_rng(100+seed)
    P1=O.G1.gen_seq(k0,d,n,threads=16); P2=O.G2.gen_seq(d,k0,n,threads=16)
    s=int.from_bytes(rng.bytes(32),"little")%R if seed%4 else [0,1,R-1,0xD201000000010000**2,2**64][seed//4]
    m1=ops.mul_add(ca.G1,P1,s,P1[::-1].copy() if seed%2 else None); m2=ops.mul_add(ca.G2,P2,s,P2[::-1].copy() if seed%2 else None)
    for i in range(0,n,3):
        for G,Pp,m,tag in ((O.G1,P1,m1,"same1"),(O.G2,P2,m2,"same2")):
            e=G.mul(Pp[i],O.int_to_limbs(s,4))
            if seed%2: e=G.add(e,G.mul(Pp[n-1-i],O.int_to_limbs(1,4)))
            e=G.to_affine(e)
            if not (e[1] and not m[i].any() or (m[i]==e[0]).all()): bad+=1; print(tag,seed,i)
print("fuzz_scale mismatches:",bad)

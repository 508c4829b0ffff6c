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
# the same step with the doubling chains done ahead of the scalar (dgpu_fold_prepare_pair / dgpu_g*_fold_apply): random and structured scalars, several per table,
# against the chain kernels above
import ctypes as C
from crypto_amd._native import lib
p=lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
X=0xD201000000010000; LAM=X*X-1
for seed in range(int(sys.argv[1]) if len(sys.argv)>1 else 12):
    rng=np.random.default_rng(500+seed)
    n1=int(rng.integers(1,700)); n2=int(rng.integers(1,400))
    P1=np.ascontiguousarray(O.G1.gen_seq(k0,d,n1,threads=16)); P2=np.ascontiguousarray(O.G2.gen_seq(d,k0,n2,threads=16))
    if n1>3: P1[int(rng.integers(0,n1))]=0
    if n2>3: P2[int(rng.integers(0,n2))]=0
    h1,h2=C.c_uint64(0),C.c_uint64(0)
    assert lib().dgpu_fold_prepare_pair(p(P1),n1,C.byref(h1),p(P2),n2,C.byref(h2))==0
    for rep in range(4):
        kind=int(rng.integers(0,4))
        if kind==0: s=int.from_bytes(rng.bytes(32),"little")%R
        elif kind==1:
            dg=[int(rng.choice([0,1,X-1,int(rng.integers(0,2**63))])) for _ in range(4)]; dg[3]%=R//X**3; s=(dg[0]+dg[1]*X+dg[2]*X*X+dg[3]*X**3)%R
        elif kind==2: s=(int(rng.choice([0,1,2**127,2**128-1]))+int(rng.choice([0,1,2**126,LAM-1]))*LAM)%R
        else: s=int(rng.choice([0,1,2,R-1]))
        A1=None if rep==3 else P1[::-1].copy(); A2=None if rep==3 else P2[::-1].copy()
        sl=ops.limbs([s]).reshape(4)
        o1=np.zeros_like(P1); i1=np.zeros(n1,np.uint8); o2=np.zeros_like(P2); i2=np.zeros(n2,np.uint8)
        assert lib().dgpu_g1_fold_apply(h1.value,p(sl),p(A1),p(o1),p(i1))==0 and lib().dgpu_g2_fold_apply(h2.value,p(sl),p(A2),p(o2),p(i2))==0
        w1=ops.mul_add(ca.G1,P1,s,A1); w2=ops.mul_add(ca.G2,P2,s,A2)
        if not (o1==w1).all(): bad+=1; print("fold1",seed,rep,hex(s))
        if not (o2==w2).all(): bad+=1; print("fold2",seed,rep,hex(s))
    lib().dgpu_fold_free(h1.value); lib().dgpu_fold_free(h2.value)
print("fuzz_scale mismatches:",bad)

import sys, time
sys.path.insert(0,"/root/repo"); sys.path.insert(0,"/root/repo/oracle")
import crypto_amd as ca, oracle_c as O
from crypto_amd import fixed_base as FB
ca.init(0)
_twin = ca.twin(); _twin.__enter__()      # knobs / stage timers live in the development twin (include/dock_gpu_dev.h): this script runs on it
for lg in (12, 14, 16, 18, 20):
    n=1<<lg
    with FB.WindowTable(ca.G2, O.G2.generator()) as t2:
        db=t2.multiply_many_to_bases(O.rand_scalars(3,n))
    ds=ca.DeviceScalars(O.rand_scalars(4,n))
    for _ in range(8): db.msm_resident(ds)
    t0=time.time()
    for _ in range(10): db.msm_resident(ds)
    dt=(time.time()-t0)/10
    ca.prof.enable(True); ca.prof.reset()
    for _ in range(4): db.msm_resident(ds)
    st=ca.prof.read(); ca.prof.enable(False)
    print("G2 2^%d %.3f ms |"%(lg, dt*1e3), " ".join("%s=%.3f"%(k.split(".")[1],v[0]/v[1]) for k,v in st.items()), flush=True)
    db.free(); ds.free()

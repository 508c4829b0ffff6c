# tools/dev/ws_chain_time.py — the line kernel ALONE (G2Prepared::from runs the whole 68-step chain in one launch with nothing beside it): run under
# rocprofv3 --kernel-trace --stats with MLMODE=7 / 15 and N=3 / 1024 to read the chain's duration per form
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT + "/oracle", ROOT]
import oracle_c as O, crypto_amd as ca
from crypto_amd import pairing
from crypto_amd._native import lib
ca.init(0)
_tw = ca.twin(); _tw.__enter__(); assert lib().dgpu_set_miller_pipeline(int(os.environ.get("MLMODE", "15"))) == 0
n = int(os.environ.get("N", "1024"))
k0 = O.rand_scalars(41, 1)[0]; d = O.rand_scalars(42, 1)[0]
qs = O.G2.gen_seq(d, k0, n, threads=32)
for _ in range(12): pairing.G2Prepared.from_affine(qs)

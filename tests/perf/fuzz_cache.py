"""GPU fuzz of the resident-bases cache (crypto_amd/csrc/bases_cache.hpp): random sequences of one-shot MSM calls on sub-slices of a few key buffers, interleaved
with refills, announced in-place edits, unannounced edits under the exact mode, clears, budget changes and concurrent callers — every answer compared with the CPU
oracle.  Usage: SECONDS=120 SEED=1 python tests/perf/fuzz_cache.py"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, ROOT + "/oracle", ROOT + "/tests"]
import numpy as np
import torch  # noqa: F401
import oracle_c as O, util as U, crypto_amd as ca

ca.init(0)
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
T_END = time.time() + float(os.environ.get("SECONDS", "60"))
ca.bases_cache_clear(); ca.bases_cache(min_n=1 << 11, verify=24)
NK, NMAX = 4, 30000
pool, _, _ = U.seq_bases(O.G1, 4 * NMAX, 77, threads=32)                 # a pool of valid points to refill / edit from
keys = []
for k in range(NK):
    b = pool[rng.integers(0, len(pool), NMAX)].copy()
    inf = (rng.integers(0, 50, NMAX) == 0).astype(np.uint8)
    keys.append({"packed": b, "inf": inf, "st": ca.to_affine_structs(ca.G1, b, inf)})
exact = False
calls = checks = 0
lock = threading.Lock()


def sync_struct(k, lo, hi):
    K = keys[k]
    K["st"]["x"][lo:hi], K["st"]["y"][lo:hi], K["st"]["infinity"][lo:hi] = K["packed"][lo:hi, :6], K["packed"][lo:hi, 6:], K["inf"][lo:hi]


def one_call(k, lo, hi, strided, r):
    global calls
    K = keys[k]
    sc = O.rand_scalars(int(r.integers(1, 1 << 30)), hi - lo)
    if strided:
        got = ca.msm_strided(ca.G1, K["st"][lo:hi], sc)
    else:
        got = ca.msm_bigint(ca.G1, K["packed"][lo:hi], sc, K["inf"][lo:hi])
    keep = K["inf"][lo:hi] == 0
    ref = O.G1.msm(K["packed"][lo:hi][keep], sc[keep], threads=8)
    assert U.jac_to_model(O.G1, got) == U.jac_to_model(O.G1, ref), ("MISMATCH", k, lo, hi, strided, exact, ca.bases_cache_stats())
    with lock:
        calls += 1


while time.time() < T_END:
    op = rng.integers(0, 100)
    k = int(rng.integers(0, NK))
    if op < 70:                                   # a call on a slice (often the same few shapes, so that entries become resident and get hit)
        shape = int(rng.integers(0, 4))
        lo, hi = [(0, NMAX), (1, NMAX), (0, 20000), (int(rng.integers(0, 9000)), int(rng.integers(12000, NMAX)))][shape]
        one_call(k, lo, hi, bool(rng.integers(0, 2)), rng)
    elif op < 76:                                 # the buffer refilled with another key (no announcement: the sampled check must notice)
        keys[k]["packed"][:] = pool[rng.integers(0, len(pool), NMAX)]
        sync_struct(k, 0, NMAX)
    elif op < 84:                                 # an in-place edit of one record: announced under the sampled default, silent under the exact mode
        i = int(rng.integers(0, NMAX))
        keys[k]["packed"][i] = pool[rng.integers(0, len(pool))]
        if rng.integers(0, 3) == 0:
            keys[k]["inf"][i] ^= 1
        sync_struct(k, i, i + 1)
        if not exact:
            ca.bases_cache_invalidate(keys[k]["packed"][i:i + 1]); ca.bases_cache_invalidate(keys[k]["st"][i:i + 1]); ca.bases_cache_invalidate(keys[k]["inf"][i:i + 1])
    elif op < 88:
        exact = not exact
        if not exact:
            ca.bases_cache_clear()                # (silent edits made under the exact mode were never announced: leaving that mode means starting over)
        ca.bases_cache(verify=ca.CACHE_VERIFY_FULL if exact else 24)
    elif op < 91:
        ca.bases_cache_clear()
    elif op < 94:
        ca.bases_cache(bytes=int(rng.choice([40 << 20, 150 << 20, (1 << 64) - 1])))
    else:                                         # six concurrent callers on their own slices of one key
        th = [threading.Thread(target=one_call, args=(k, 0 if t % 2 else 1, NMAX, bool(t % 3), np.random.default_rng(int(rng.integers(1, 1 << 30))))) for t in range(6)]
        [t.start() for t in th]; [t.join() for t in th]
    checks += 1
print("fuzz_cache ok: %d operations, %d MSM calls compared with the oracle, final stats %s" % (checks, calls, ca.bases_cache_stats()), flush=True)

"""GPU (-m gpu): the resident-bases cache behind the one-shot MSM entry points (include/dock_gpu.h `dgpu_set_bases_cache_*`, crypto_amd/csrc/bases_cache.hpp).
The reference's call sites pass the same proving-key slices proof after proof (legogroth16/src/prover.rs:286,299,363,592) and know nothing of handles: the
unmodified `dgpu_msm_*` / `dgpu_msm_*_strided` call must reach the resident table by itself, return the same group element limb for limb whichever path
served it, and never answer from a key whose host memory has changed.  Every result is compared with the CPU oracle (or the closed form over known discrete
logs where the oracle's MSM would take too long)."""
import threading
import numpy as np
import pytest
import torch
import oracle_c as O
import util as U
import crypto_amd as ca

pytestmark = pytest.mark.gpu
CUR = {"G1": (ca.G1, O.G1), "G2": (ca.G2, O.G2)}
AUTO = (1 << 64) - 1


@pytest.fixture(autouse=True)
def _fresh_cache():
    assert torch.cuda.is_available(), "GPU tests need a device"
    ca.init(0)
    ca.bases_cache_clear()
    ca.bases_cache(bytes=AUTO, min_n=1 << 12, verify=24)
    yield
    ca.bases_cache_clear()
    ca.bases_cache(bytes=AUTO, min_n=1 << 16, verify=ca.CACHE_VERIFY_FULL)       # (the library's defaults)


def delta(before):
    now = ca.bases_cache_stats()
    return {k: now[k] - before[k] for k in ("hits", "misses", "fills", "stale", "evictions")}


def oracle_point(G, bases, sc, inf=None):
    keep = slice(None) if inf is None else (np.asarray(inf) == 0)
    return U.jac_to_model(G, G.msm(bases[keep], sc[keep], threads=16))


@pytest.mark.parametrize("gname,n", [("G1", 5000), ("G1", 40000), ("G2", 5000), ("G2", 33000)])
def test_first_call_one_shot_second_fills_third_hits(gname, n):
    """n = 5000: the entry stays a plain handle (too few points for a bucket table); 40000 / 33000: a width-16 table"""
    curve, G = CUR[gname]
    bases, _, _ = U.seq_bases(G, n, 300 + n, threads=16)
    inf = np.zeros(n, np.uint8); inf[5::97] = 1
    st = ca.to_affine_structs(curve, bases, inf)
    s0 = ca.bases_cache_stats()
    for call in range(4):
        sc = O.rand_scalars(900 + call, n)
        got = ca.msm_strided(curve, st, sc)
        assert U.jac_to_model(G, got) == oracle_point(G, bases, sc, inf), (gname, n, call)
        # the Montgomery form takes the same entry
        if call == 3:
            assert (ca.msm_strided(curve, st, O.fr_to_mont(sc), montgomery=True) == got).all()
    d = delta(s0)
    assert d == {"hits": 4, "misses": 1, "fills": 1, "stale": 0, "evictions": 0}, d
    assert ca.bases_cache_stats()["entries"] == 1


def test_packed_layout_with_separate_flags_is_cached_too():
    curve, G = CUR["G1"]
    n = 20000
    bases, _, _ = U.seq_bases(G, n, 41, threads=16)
    inf = np.zeros(n, np.uint8); inf[::11] = 1
    s0 = ca.bases_cache_stats()
    outs = []
    for call in range(3):
        sc = O.rand_scalars(50 + call, n)
        outs.append((ca.msm_bigint(curve, bases, sc, inf), oracle_point(G, bases, sc, inf)))
    assert all(U.jac_to_model(G, g) == r for g, r in outs)
    assert delta(s0)["hits"] == 2 and delta(s0)["fills"] == 1


def test_sub_slices_resolve_to_the_resident_entry():
    """`&query[1..]` (legogroth16/src/prover.rs:592) and a truncated length (prover.rs:286) hit the entry of the whole slice"""
    curve, G = CUR["G1"]
    n = 36000
    bases, _, _ = U.seq_bases(G, n, 61, threads=16)
    st = ca.to_affine_structs(curve, bases)
    sc = O.rand_scalars(62, n)
    ca.msm_strided(curve, st, sc); ca.msm_strided(curve, st, sc)          # resident now
    s0 = ca.bases_cache_stats()
    assert U.jac_to_model(G, ca.msm_strided(curve, st[1:], sc[:n - 1])) == oracle_point(G, bases[1:], sc[:n - 1])
    assert U.jac_to_model(G, ca.msm_strided(curve, st[:30000], sc)) == oracle_point(G, bases[:30000], sc[:30000])
    assert U.jac_to_model(G, ca.msm_strided(curve, st[777:20001], sc)) == oracle_point(G, bases[777:20001], sc[:20001 - 777])
    assert delta(s0) == {"hits": 3, "misses": 0, "fills": 0, "stale": 0, "evictions": 0}
    # the packed layout of the same points is another layout: not confused with the structs
    assert U.jac_to_model(G, ca.msm_bigint(curve, bases, sc)) == oracle_point(G, bases, sc)
    assert delta(s0)["misses"] == 1


def test_a_key_whose_host_memory_changed_is_never_used():
    curve, G = CUR["G1"]
    n = 20000
    b1, _, _ = U.seq_bases(G, n, 71, threads=16)
    b2, _, _ = U.seq_bases(G, n, 73, threads=16)
    sc = O.rand_scalars(74, n)
    st = ca.to_affine_structs(curve, b1)
    r1, r2 = oracle_point(G, b1, sc), oracle_point(G, b2, sc)
    for _ in range(3):
        assert U.jac_to_model(G, ca.msm_strided(curve, st, sc)) == r1
    # (a) the buffer is refilled with another key: noticed at once, the answer is the new key's
    s0 = ca.bases_cache_stats()
    st["x"], st["y"] = b2[:, :6], b2[:, 6:]
    assert U.jac_to_model(G, ca.msm_strided(curve, st, sc)) == r2
    assert delta(s0)["stale"] == 1 and delta(s0)["hits"] == 0
    # ... and the new contents go through the same sequence: noted, then resident
    for _ in range(3):
        assert U.jac_to_model(G, ca.msm_strided(curve, st, sc)) == r2
    assert delta(s0)["fills"] == 1 and delta(s0)["hits"] == 3      # (the stale call itself was the new contents' first sighting)
    # (b) the first record alone changes (always among the samples)
    st["x"][0], st["y"][0] = b1[0, :6], b1[0, 6:]
    b3 = b2.copy(); b3[0] = b1[0]
    s0 = ca.bases_cache_stats()
    assert U.jac_to_model(G, ca.msm_strided(curve, st, sc)) == oracle_point(G, b3, sc)
    assert delta(s0)["stale"] == 1
    # (c) one record in the middle: the exact mode (the library's default; this file's fixture selects the sampled one) re-fingerprints every record of every call
    ca.bases_cache(verify=ca.CACHE_VERIFY_FULL)
    for _ in range(3):
        assert U.jac_to_model(G, ca.msm_strided(curve, st, sc)) == oracle_point(G, b3, sc)
    st["x"][12345], st["y"][12345] = b1[777, :6], b1[777, 6:]
    b4 = b3.copy(); b4[12345] = b1[777]
    s0 = ca.bases_cache_stats()
    assert U.jac_to_model(G, ca.msm_strided(curve, st, sc)) == oracle_point(G, b4, sc)
    assert delta(s0)["stale"] == 1
    # (d) the same edit under the sampled mode, announced by the host
    ca.bases_cache(verify=24)
    for _ in range(3):
        ca.msm_strided(curve, st, sc)
    st["x"][4321], st["y"][4321] = b1[5, :6], b1[5, 6:]
    b5 = b4.copy(); b5[4321] = b1[5]
    ca.bases_cache_invalidate(st)
    assert U.jac_to_model(G, ca.msm_strided(curve, st, sc)) == oracle_point(G, b5, sc)
    # (e) the identity flag inside the struct is part of a record's fingerprint
    for _ in range(3):
        ca.msm_strided(curve, st, sc)
    st["infinity"][0] = 1
    inf = np.zeros(n, np.uint8); inf[0] = 1
    assert U.jac_to_model(G, ca.msm_strided(curve, st, sc)) == oracle_point(G, b5, sc, inf)


def test_least_recently_used_entries_leave_under_the_budget():
    curve, G = CUR["G1"]
    n = 40000                                   # a width-16 table: 16 rows x 128 B x n = 82 MB
    keys = []
    for k in range(3):
        b, _, _ = U.seq_bases(G, n, 81 + 2 * k, threads=16)
        keys.append((b, ca.to_affine_structs(curve, b)))
    sc = O.rand_scalars(89, n)
    refs = [oracle_point(G, b, sc) for b, _ in keys]
    ca.bases_cache(bytes=200 << 20)             # room for two of them
    s0 = ca.bases_cache_stats()
    for rnd in range(3):
        for k, (b, st) in enumerate(keys):
            assert U.jac_to_model(G, ca.msm_strided(curve, st, sc)) == refs[k], (rnd, k)
    d = delta(s0)
    assert d["evictions"] >= 1 and d["fills"] >= 3, d
    now = ca.bases_cache_stats()
    assert now["entries"] <= 2 and now["bytes"] <= 200 << 20
    # a budget too small for any table: nothing is cached, every call is one-shot
    ca.bases_cache_clear(); ca.bases_cache(bytes=1 << 20)
    s0 = ca.bases_cache_stats()
    for _ in range(3):
        assert U.jac_to_model(G, ca.msm_strided(curve, keys[0][1], sc)) == refs[0]
    assert delta(s0)["fills"] == 0 and ca.bases_cache_stats()["entries"] == 0
    # off
    ca.bases_cache(bytes=0)
    s0 = ca.bases_cache_stats()
    for _ in range(3):
        assert U.jac_to_model(G, ca.msm_strided(curve, keys[0][1], sc)) == refs[0]
    assert delta(s0) == {"hits": 0, "misses": 0, "fills": 0, "stale": 0, "evictions": 0}


def test_six_threads_on_one_key_from_cold():
    """rayon workers of one prover (verifiable_encryption/src/tz_21/rdkgith.rs:140-147): the entry is filled by one of them, the others run one-shot meanwhile"""
    curve, G = CUR["G1"]
    n = 34000
    bases, _, _ = U.seq_bases(G, n, 95, threads=16)
    st = ca.to_affine_structs(curve, bases)
    scs = [O.rand_scalars(960 + t, n) for t in range(6)]
    refs = [oracle_point(G, bases, s) for s in scs]
    s0 = ca.bases_cache_stats()
    bad = []

    def work(t):
        for rnd in range(4):
            if U.jac_to_model(G, ca.msm_strided(curve, st, scs[t])) != refs[t]:
                bad.append((t, rnd))
    th = [threading.Thread(target=work, args=(t,)) for t in range(6)]
    [x.start() for x in th]; [x.join() for x in th]
    assert not bad, bad
    d = delta(s0)
    assert d["fills"] == 1 and d["hits"] >= 1 and d["hits"] + d["misses"] == 24, d      # (the other threads run one-shot while one of them fills the entry)
    s0 = ca.bases_cache_stats()
    th = [threading.Thread(target=work, args=(t,)) for t in range(6)]
    [x.start() for x in th]; [x.join() for x in th]
    assert not bad and delta(s0) == {"hits": 24, "misses": 0, "fills": 0, "stale": 0, "evictions": 0}
    # a clear while calls are in flight: entries in use are released by their last user
    stop = []

    def spin(t):
        while not stop:
            if U.jac_to_model(G, ca.msm_strided(curve, st, scs[t])) != refs[t]:
                bad.append(("spin", t))
    th = [threading.Thread(target=spin, args=(t,)) for t in range(3)]
    [x.start() for x in th]
    for _ in range(10):
        ca.bases_cache_clear()
        ca.msm_strided(curve, st, scs[5])
    stop.append(1); [x.join() for x in th]
    assert not bad, bad


def test_full_size_key_through_the_unmodified_call():
    """BASELINE config 2's shape through dgpu_msm_g1_strided: cold, fill, warm — the closed form over known discrete logs at 2^20 terms"""
    curve, G = CUR["G1"]
    ca.bases_cache(min_n=1 << 16)
    n = 1 << 20
    bases, k0, d = U.seq_bases(G, n, 111, threads=16)
    st = ca.to_affine_structs(curve, bases)
    s0 = ca.bases_cache_stats()
    for call in range(3):
        sc = O.rand_scalars(120 + call, n)
        assert U.jac_to_model(G, ca.msm_strided(curve, st, sc)) == U.closed_form(G, sc, k0, d)
    sc = O.rand_scalars(130, n - 1)
    assert U.jac_to_model(G, ca.msm_strided(curve, st[1:], sc)) == U.closed_form(G, sc, k0 + d, d)      # `&query[1..]`
    assert delta(s0) == {"hits": 3, "misses": 1, "fills": 1, "stale": 0, "evictions": 0}

"""Host-side tile schedule of the 256-wide persistent GEMM (bf_gemm_schedule, csrc/bf_gemm256.hip): every 32-row unit
of every (sample, layer, n-tile) column is covered exactly once, no tile is taller than 8 units, and the units are
spread evenly over the workgroups — the property that removes the partial last round of fixed 256-row tiles
(BERT-base: 480 k tiles on 256 CUs = 1.875 k rounds).  CPU-only: the schedule is plain host code."""
import ctypes

import numpy as np
import pytest

from bayeformers_amd import _C


def schedule(S, L, M, N, n_cu=256):
    lib = _C.lib()
    r, g = ctypes.c_int(), ctypes.c_int()
    n = lib.bf_gemm_schedule(S, L, M, N, n_cu, None, 0, ctypes.byref(r), ctypes.byref(g))
    out = np.zeros(n, dtype=np.int32)
    assert lib.bf_gemm_schedule(S, L, M, N, n_cu, out.ctypes.data, n, ctypes.byref(r), ctypes.byref(g)) == n
    return out.reshape(r.value, g.value, 4)


CASES = [(10, 1, 4096, 768), (10, 3, 4096, 768), (10, 1, 4096, 3072), (10, 1, 6144, 1024), (10, 1, 6144, 4096),
         (8, 1, 4096, 768), (64, 1, 4096, 768), (1, 1, 4096, 4096), (4, 1, 256, 128), (10, 1, 4100, 768),
         (3, 1, 130, 192), (1, 1, 65, 256), (2, 2, 1000, 520), (7, 1, 33, 768)]


@pytest.mark.parametrize("S,L,M,N", CASES)
def test_schedule_covers_every_unit_once(S, L, M, N):
    t = schedule(S, L, M, N)
    rounds, grid, _ = t.shape
    tiles_n, units = (N + 255) // 256, (M + 31) // 32
    seen = np.zeros((S * L, tiles_n, units), dtype=np.int32)
    for j in range(rounds):
        for b in range(grid):
            pair, xs, z, m0 = (int(v) for v in t[j, b])
            h, tn = z >> 24, z & 0xFFFFFF
            if h == 0:
                assert not t[j:, b, 2].any(), "a workgroup's list ends at its first empty entry"
                continue
            assert 1 <= h <= 8 and m0 % 32 == 0 and 0 <= tn < tiles_n
            assert xs == pair % S and 0 <= pair < S * L
            seen[pair, tn, m0 // 32:m0 // 32 + h] += 1
            assert m0 // 32 + h <= units
    assert (seen == 1).all()


@pytest.mark.parametrize("S,L,M,N", [c for c in CASES if c[2] >= 1024])
def test_schedule_is_balanced(S, L, M, N):
    t = schedule(S, L, M, N)
    load = (t[:, :, 2] >> 24).sum(0)
    total = S * L * ((N + 255) // 256) * ((M + 31) // 32)
    assert load.sum() == total
    # every workgroup within one tile-height step of the mean (fixed 256-row tiles leave up to 8 units of slack)
    assert load.max() - load.min() <= 1, (load.min(), load.max())


def test_bert_base_launches_have_no_partial_round():
    for L, N in ((1, 768), (3, 768), (1, 3072)):
        t = schedule(10, L, 4096, N)
        load = (t[:, :, 2] >> 24).sum(0)
        assert load.min() == load.max() == 10 * L * ((N + 255) // 256) * 128 // 256


def schedule_policy(S, L, M, N, policy, n_cu=256):
    lib = _C.lib()
    r, g = ctypes.c_int(), ctypes.c_int()
    n = lib.bf_gemm_schedule_policy(S, L, M, N, n_cu, policy, None, 0, ctypes.byref(r), ctypes.byref(g))
    out = np.zeros(n, dtype=np.int32)
    assert lib.bf_gemm_schedule_policy(S, L, M, N, n_cu, policy, out.ctypes.data, n, ctypes.byref(r), ctypes.byref(g)) == n
    return out.reshape(r.value, g.value, 4)


def fetch_rows(t):
    flat = np.ascontiguousarray(t.reshape(-1), dtype=np.int32)
    return _C.lib().bf_gemm_schedule_fetch_rows(flat.ctypes.data, t.shape[0], t.shape[1])


def test_fetch_model_counts_distinct_panels_per_xcd_round():
    # one round, 16 workgroups = 2 per XCD: the workgroups b and b + 8 share a W panel (same pair and n-tile) and 4 of their 8 x units
    t = np.zeros((1, 16, 4), dtype=np.int32)
    for b in range(16):
        t[0, b] = (b % 8, b % 8, 0 | (8 << 24), 0 if b < 8 else 4 * 32)
    assert fetch_rows(t) == 8 * (256 + 12 * 32)
    assert _C.lib().bf_gemm_schedule_fetch_rows(None, 1, 16) == -1


@pytest.mark.parametrize("S,L,M,N,K,measured_mb", [(10, 3, 4096, 768, 768, 308), (10, 1, 4096, 768, 768, 108),
                                                     (10, 1, 4096, 3072, 768, 385), (10, 1, 4096, 768, 3072, 428)])
def test_fetch_model_reproduces_the_counted_fabric_reads_of_the_round5_policy(S, L, M, N, K, measured_mb):
    """TCC_EA0_RDREQ x 128 B per launch of the BERT-base step under policy 12 (profiles/r6a_pmc_gemm_positions.md)."""
    mb = fetch_rows(schedule_policy(S, L, M, N, 12)) * K * 2 / 1e6
    assert abs(mb - measured_mb) <= 0.015 * measured_mb, mb


@pytest.mark.parametrize("S,L,M,N", [c for c in CASES if c[2] >= 1024])
def test_default_policy_keeps_the_tiles_and_never_fetches_more(S, L, M, N):
    old, new = schedule_policy(S, L, M, N, 12), schedule(S, L, M, N)
    assert sorted((old[:, :, 2] >> 24).sum(0)) == sorted((new[:, :, 2] >> 24).sum(0))  # same units per workgroup
    assert fetch_rows(new) <= fetch_rows(old) * 1.02

"""How many operand bytes does the 256-wide GEMM's tile schedule pull into the eight private L2s?  A model, not a
measurement: every XCD's 32 workgroups run their j-th tiles together; an XCD's L2 is an LRU over x row-units
((sample, 32-row unit): 32 K e bytes) and W panels ((pair, n-tile): 256 K e bytes) of `--l2` MiB.  Reports modelled fetch
bytes against the algorithmic operand bytes, for the library's schedule (bf_gemm_schedule) of a launch.

    python tools/sched_l2_sim.py [S L M N K] ...        (defaults: the BERT-base launches)
"""
import collections
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from bayeformers_amd import _C  # noqa: E402


def schedule(S, L, M, N, n_cu=256):
    lib = _C.lib()
    rounds, grid = ctypes.c_int(), ctypes.c_int()
    n = lib.bf_gemm_schedule(S, L, M, N, n_cu, None, 0, ctypes.byref(rounds), ctypes.byref(grid))
    buf = np.zeros(n, dtype=np.int32)
    lib.bf_gemm_schedule(S, L, M, N, n_cu, buf.ctypes.data, n, ctypes.byref(rounds), ctypes.byref(grid))
    return buf.reshape(rounds.value, grid.value, 4), rounds.value, grid.value


def simulate(tab, K, l2_mib, es=2):
    rounds, grid, _ = tab.shape
    cap = l2_mib * (1 << 20)
    fetched = 0
    for xcd in range(8):
        lru = collections.OrderedDict()
        used = 0
        for j in range(rounds):
            want = []
            for b in range(xcd, grid, 8):  # block b runs on XCD b % 8
                pair, xs, z, m0 = (int(v) for v in tab[j, b])
                h = z >> 24
                if h == 0:
                    continue
                want.append((("w", pair, z & 0xFFFFFF), 256 * K * es))
                for u in range(m0 // 32, m0 // 32 + h):
                    want.append((("x", xs, u), 32 * K * es))
            for key, size in want:
                if key in lru:
                    lru.move_to_end(key)
                    continue
                fetched += size
                lru[key] = size
                used += size
                while used > cap:
                    _, sz = lru.popitem(last=False)
                    used -= sz
    return fetched


def main():
    v = [int(a) for a in sys.argv[1:] if not a.startswith("--")]
    l2 = 4
    shapes = [tuple(v[i:i + 5]) for i in range(0, len(v), 5)] or [
        (10, 3, 4096, 768, 768), (10, 1, 4096, 768, 768), (10, 1, 4096, 3072, 768), (10, 1, 4096, 768, 3072)]
    for S, L, M, N, K in shapes:
        tab, rounds, grid = schedule(S, L, M, N)
        alg = S * (M * K + L * N * K) * 2
        f = simulate(tab, K, l2)
        print(f"S={S} L={L} M={M} N={N} K={K}: {rounds} rounds x {grid} workgroups; modelled L2 fills {f / 1e6:.0f} MB "
              f"= {f / alg:.2f} x the {alg / 1e6:.0f} MB of operands")


if __name__ == "__main__":
    main()

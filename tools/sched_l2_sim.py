"""How many operand bytes does the 256-wide GEMM's tile schedule pull into the eight private L2s?  A model — validated against
TCC_EA0_RDREQ to 1 % on the four BERT-base launches (profiles/r6b_sched_l2_model.md): every XCD's 32 workgroups run their j-th
tiles together and sweep k in lockstep; an XCD's L2 is an LRU over k-slices (one 64-element k-step of a 32-row x unit or of a
256-row W panel) of `--l2` MiB.  The library's own closed form of the same model (distinct panels per XCD and round: nothing
survives from one round to the next) is bf_gemm_schedule_fetch_rows, printed beside it.

    python tools/sched_l2_sim.py [--policy P] [--l2 MiB] [S L M N K] ...        (defaults: the BERT-base launches)
"""
import collections
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from bayeformers_amd import _C  # noqa: E402


def schedule(S, L, M, N, n_cu=256, policy=-1):
    lib = _C.lib()
    rounds, grid = ctypes.c_int(), ctypes.c_int()
    n = lib.bf_gemm_schedule_policy(S, L, M, N, n_cu, policy, None, 0, ctypes.byref(rounds), ctypes.byref(grid))
    buf = np.zeros(n, dtype=np.int32)
    lib.bf_gemm_schedule_policy(S, L, M, N, n_cu, policy, buf.ctypes.data, n, ctypes.byref(rounds), ctypes.byref(grid))
    return buf.reshape(rounds.value, grid.value, 4), rounds.value, grid.value


def model_rows(tab):
    t = np.ascontiguousarray(tab.reshape(-1), dtype=np.int32)
    return _C.lib().bf_gemm_schedule_fetch_rows(t.ctypes.data, tab.shape[0], tab.shape[1])


def simulate(tab, K, l2_mib=4.0, es=2, kstep=64):
    rounds, grid, _ = tab.shape
    cap = int(l2_mib * (1 << 20))
    x_bytes, w_bytes = 32 * kstep * es, 256 * kstep * es
    fetched = 0
    for xcd in range(8):
        lru = collections.OrderedDict()
        used = 0
        for j in range(rounds):
            tiles = []
            for b in range(xcd, grid, 8):  # block b runs on XCD b % 8
                pair, xs, z, m0 = (int(v) for v in tab[j, b])
                if z >> 24:
                    tiles.append((pair, xs, z & 0xFFFFFF, m0 // 32, z >> 24))
            for k in range(K // kstep):
                for pair, xs, tn, u0, h in tiles:
                    want = [(("w", pair, tn, k), w_bytes)] + [(("x", xs, u, k), x_bytes) for u in range(u0, u0 + h)]
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
    argv = sys.argv[1:]
    policy, l2 = -1, 4.0
    if "--policy" in argv:
        i = argv.index("--policy")
        policy = int(argv[i + 1], 0)
        del argv[i:i + 2]
    if "--l2" in argv:
        i = argv.index("--l2")
        l2 = float(argv[i + 1])
        del argv[i:i + 2]
    v = [int(a) for a in argv]
    shapes = [tuple(v[i:i + 5]) for i in range(0, len(v), 5)] or [
        (10, 3, 4096, 768, 768), (10, 1, 4096, 768, 768), (10, 1, 4096, 3072, 768), (10, 1, 4096, 768, 3072)]
    for S, L, M, N, K in shapes:
        tab, rounds, grid = schedule(S, L, M, N, policy=policy)
        alg = S * (M * K + L * N * K) * 2
        f = simulate(tab, K, l2)
        c = model_rows(tab) * K * 2
        print(f"S={S} L={L} M={M} N={N} K={K}: {rounds} rounds x {grid} workgroups; L2 fills, LRU over k-slices {f / 1e6:.0f} MB "
              f"= {f / alg:.2f} x the {alg / 1e6:.0f} MB of operands; closed form {c / 1e6:.0f} MB = {c / alg:.2f} x")


if __name__ == "__main__":
    main()

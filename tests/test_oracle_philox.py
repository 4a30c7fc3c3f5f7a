"""The oracle's epsilon generator: Random123 known-answer vectors, C vs numpy, committed KAT, host twin."""
import numpy as np
import pytest

from oracle import bayes_oracle as bo

# Random123 kat_vectors, `philox4x32 R counter[4] key[2] -> output[4]`, for the contract's R = 7 and for R = 10
# (the same round function, so the second set pins it independently of the round count)
PI_CTR, PI_KEY = [0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]
KAT = {
    7: [([0, 0, 0, 0], [0, 0], [0x5F6FB709, 0x0D893F64, 0x4F121F81, 0x4F730A48]),
        ([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2, [0x5207DDC2, 0x45165E59, 0x4D8EE751, 0x8C52F662]),
        (PI_CTR, PI_KEY, [0x4DFCCABA, 0x190A87F0, 0xC47362BA, 0xB6B5242A])],
    10: [([0, 0, 0, 0], [0, 0], [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]),
         ([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2, [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]),
         (PI_CTR, PI_KEY, [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1])],
}


def test_philox_known_answers():
    assert bo.PHILOX_ROUNDS == 7
    for rounds, vectors in KAT.items():
        for ctr, key, exp in vectors:
            assert list(bo.philox4x32(ctr, key, rounds)) == exp
            assert list(bo.philox4x32_numpy(np.array(ctr), key, rounds)) == exp
    ctr, key, exp = KAT[7][2]
    assert list(bo.philox4x32(ctr, key)) == exp  # the default round count is the contract's


def test_c_and_numpy_normals_agree_bitwise():
    for n, seed, sample, stream, off in [(1000, 0x5EED, 3, 7, 5), (17, 2**63 + 11, 2**32 - 1, 147, 0), (1, 1, 0, 0, 3)]:
        a = bo.normals(n, seed, sample, stream, off)
        b = bo.normals_numpy(n, seed, sample, stream, off)
        assert np.array_equal(a, b)


def test_offset_is_a_pure_index():
    full = bo.normals(64, 9, 2, 4)
    for off in (1, 2, 3, 4, 13):
        assert np.array_equal(bo.normals(20, 9, 2, 4, off), full[off:off + 20])


def test_committed_eps_vectors(golden_dir):
    g = np.load(f"{golden_dir}/eps_kat.npz")
    seed = int(g["seed"])
    assert np.array_equal(bo.normals(64, seed, 0, 0), g["z_s0_str0"])
    assert np.array_equal(bo.normals(64, seed, 9, 5, 3), g["z_s9_str5_off3"])


def test_moments():
    z = bo.normals(1 << 20, 123, 0, 0).astype(np.float64)
    assert abs(z.mean()) < 4e-3 and abs(z.std() - 1) < 3e-3
    assert abs((z ** 3).mean()) < 2e-2 and abs((z ** 4).mean() - 3) < 5e-2
    assert np.abs(z).max() < 6.8  # sqrt(-2 ln 2^-33)


def test_product_host_twin_matches_oracle():
    """bf_philox_normal_host (product library, host code only) against the independent oracle."""
    from bayeformers_amd import ops

    for n, seed, sample, stream, off in [(4099, 0x5EED, 0, 0, 0), (257, 77, 12, 147, 6)]:
        a = ops.philox_normal_host(n, seed, sample, stream, off).numpy()
        assert np.array_equal(a, bo.normals(n, seed, sample, stream, off))


def test_dropout_host_twin_matches_the_oracle_restatement():
    """bf_dropout_keep_host (the library's host twin of the kernels' keep decisions) against the oracle's independent
    numpy restatement of the dropout contract, incl. groups past 2^32 and the rate the 16-bit threshold realises."""
    import torch

    from bayeformers_amd import ops
    from oracle import bayes_oracle as bo

    for p, seed, call, site, first in [(0.1, 0x5EED, 0, 1, 0), (0.1, 0x5EED, 7, 3, 2 ** 32 - 5), (0.5, 12345678901234, 9, 250, 17),
                                       (0.0, 1, 2, 3, 0)]:
        d = ops.Dropout(p, seed, call, site)
        host = ops.dropout_keep_host(first, 4096, d).numpy()
        ref = bo.dropout_keep(first, 4096, p, seed, call, site)
        assert np.array_equal(host, ref)
        assert abs(host.mean() - (1.0 - p)) < 0.01
        assert d.keep_scale == pytest.approx(bo.dropout_keep_scale(p), rel=1e-12)
    # different call / site / seed -> different masks
    a = bo.dropout_keep(0, 512, 0.1, 1, 0, 1)
    assert not np.array_equal(a, bo.dropout_keep(0, 512, 0.1, 1, 1, 1)) and not np.array_equal(a, bo.dropout_keep(0, 512, 0.1, 1, 0, 2))


@pytest.mark.parametrize("rounds", [7, 10])
def test_contract_header_builds_with_either_round_count(tmp_path, rounds):
    """csrc/bf_philox.h is the epsilon contract shared by host and device code.  Its round count is a build option
    (-DBF_PHILOX_ROUNDS): the shipped 7 and Random123's default 10 both reproduce the Random123 known-answer vectors from
    the header's own block function, and the host normal generator agrees with the oracle at that round count."""
    import os
    import shutil
    import subprocess

    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "kat.cpp"
    src.write_text("""
#include <stdio.h>
#include <stdlib.h>
#include "bf_philox.h"
int main(int argc, char** argv) {
    uint32_t v[6];
    for (int i = 0; i < 6; ++i) v[i] = (uint32_t)strtoul(argv[1 + i], 0, 16);
    const bf_u32x4 o = bf_philox4x32(v[0], v[1], v[2], v[3], v[4], v[5]);
    printf("%08x %08x %08x %08x\\n", o.x, o.y, o.z, o.w);
    float z[4];
    bf_normal4_host(5, 3, 2, 0x5EEDull, z);
    printf("%.9g %.9g %.9g %.9g\\n", z[0], z[1], z[2], z[3]);
    return 0;
}
""")
    exe = tmp_path / "kat"
    subprocess.run([gxx, "-O1", f"-DBF_PHILOX_ROUNDS={rounds}", "-I", os.path.join(root, "bayeformers_amd", "csrc"), str(src),
                    "-o", str(exe), "-lm"], check=True)
    for ctr, key, want in KAT[rounds]:
        out = subprocess.run([str(exe)] + [f"{w:x}" for w in list(ctr) + list(key)], check=True, capture_output=True, text=True)
        lines = out.stdout.splitlines()
        assert [int(w, 16) for w in lines[0].split()] == want
    # the header's host normals (group 5, sample 3, stream 2) vs the oracle's block function at the same round count
    x = bo.philox4x32([5, 3, 2, 0], [0x5EED, 0], rounds=rounds).astype(np.uint64)
    u = (x.astype(np.float32) * np.float32(2.0 ** -32) + np.float32(2.0 ** -33)).astype(np.float64)
    r0, r1 = np.sqrt(-2 * np.log(u[0])), np.sqrt(-2 * np.log(u[2]))
    want_z = [r0 * np.cos(2 * np.pi * u[1]), r0 * np.sin(2 * np.pi * u[1]), r1 * np.cos(2 * np.pi * u[3]), r1 * np.sin(2 * np.pi * u[3])]
    got_z = [float(v) for v in lines[1].split()]
    np.testing.assert_allclose(got_z, want_z, rtol=2e-6, atol=2e-7)

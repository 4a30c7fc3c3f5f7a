/* philox_oracle.c — TEST INFRASTRUCTURE ONLY (oracle).  Never linked, imported or called by the product path.
 *
 * Independent plain-C restatement of the epsilon contract used by the HIP kernels:
 *   Philox4x32-7 (Salmon, Moraes, Dror, Shaw: "Parallel Random Numbers: As Easy as 1, 2, 3", SC'11; the
 *   Random123 library's philox4x32_R with R = 7, the smallest round count the authors certify as
 *   Crush-resistant) followed by Box-Muller evaluated in fp64.  The round count is a parameter of the block function
 *   here so that the known-answer vectors of both R = 7 and R = 10 pin it.
 * It stands in for the reference's only RNG touch-point,
 *   eps = self.normal.sample(self.size)      /root/reference/bayeformers/nn/parameters/gaussian.py:100
 * which draws from torch's global generator (third-party, torch>=1.5.0 per /root/reference/requirements.txt:2);
 * the parity tests inject THIS epsilon into the real reference through that attribute
 * (tests/golden/make_golden.py), so reference and HIP path consume identical draws.
 *
 * Pinned by: the Random123 known-answer vectors (tests/test_oracle_philox.py) and the committed golden
 * fixtures generated from the real reference.
 *
 * Build: gcc -O2 -shared -fPIC -ffp-contract=off philox_oracle.c -o _build/liboracle.so -lm   (oracle/build.py)
 */
#include <math.h>
#include <stdint.h>

#define M0 0xD2511F53u
#define M1 0xCD9E8D57u
#define W0 0x9E3779B9u
#define W1 0xBB67AE85u

#define ORACLE_PHILOX_ROUNDS 7

void oracle_philox4x32(const uint32_t ctr[4], const uint32_t key[2], int rounds, uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < rounds; ++r) {
        if (r > 0) {
            k0 += W0;
            k1 += W1;
        }
        uint64_t p0 = (uint64_t)M0 * (uint64_t)c0;
        uint64_t p1 = (uint64_t)M1 * (uint64_t)c2;
        uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        uint32_t n0 = hi1 ^ c1 ^ k0;
        uint32_t n1 = lo1;
        uint32_t n2 = hi0 ^ c3 ^ k1;
        uint32_t n3 = lo0;
        c0 = n0;
        c1 = n1;
        c2 = n2;
        c3 = n3;
    }
    out[0] = c0;
    out[1] = c1;
    out[2] = c2;
    out[3] = c3;
}

/* u = x * 2^-32 + 2^-33 rounded to fp32 exactly as the device does (one cvt, one fma). */
static float unit_from_u32(uint32_t x) { return fmaf((float)x, 0x1p-32f, 0x1p-33f); }

static void box_muller(uint32_t a, uint32_t b, float* z0, float* z1) {
    double u1 = (double)unit_from_u32(a);
    double u2 = (double)unit_from_u32(b);
    double r = sqrt(-2.0 * log(u1));
    double t = 2.0 * 3.14159265358979323846264338327950288 * u2;
    *z0 = (float)(r * cos(t));
    *z1 = (float)(r * sin(t));
}

/* out[i] = eps(seed, sample, stream, element offset + i);  element e -> group e>>2, component e&3;
 * counter = {lo32(group), sample, stream, hi32(group)}, key = {lo32(seed), hi32(seed)}. */
void oracle_normals(float* out, uint64_t n, uint64_t seed, uint32_t sample, uint32_t stream, uint64_t offset) {
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint64_t i = 0;
    while (i < n) {
        uint64_t e = offset + i;
        uint64_t g = e >> 2;
        uint32_t ctr[4] = {(uint32_t)g, sample, stream, (uint32_t)(g >> 32)};
        uint32_t x[4];
        float z[4];
        oracle_philox4x32(ctr, key, ORACLE_PHILOX_ROUNDS, x);
        box_muller(x[0], x[1], &z[0], &z[1]);
        box_muller(x[2], x[3], &z[2], &z[3]);
        for (uint64_t j = e & 3; j < 4 && i < n; ++j, ++i) out[i] = z[j];
    }
}

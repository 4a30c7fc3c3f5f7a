// bf_philox.h — the epsilon contract of bayeformers_amd, shared by host and gfx950 device code.
//
// The reference draws eps with torch.distributions.Normal(0,1).sample(size) from the global generator
// (/root/reference/bayeformers/nn/parameters/gaussian.py:100).  A stateful stream cannot be regenerated
// inside a tiled kernel, sharded over GPUs or replayed in backward, so this build defines eps as a pure
// function of (seed, sample index, stream id, element index):
//
//   group g = e >> 2, component j = e & 3
//   (x0,x1,x2,x3) = Philox4x32-7(counter = {lo32(g), sample, stream, hi32(g)}, key = {lo32(seed), hi32(seed)})
//   u(x)  = fmaf((float)x, 2^-32, 2^-33)                      in (0, 1], fp32, identical on host and device
//   (z0,z1) = BoxMuller(u(x0), u(x1)), (z2,z3) = BoxMuller(u(x2), u(x3)),  eps_e = z_j
//   BoxMuller(u1,u2) = (r cos(2 pi u2), r sin(2 pi u2)),  r = sqrt(-2 ln u1)
//
// stream = 2*layer_id + tensor_id (0 = weight, 1 = bias).  The host twin evaluates Box-Muller in fp64 and
// rounds once; the device uses the gfx950 transcendental units (v_log_f32, v_sqrt_f32, v_sin_f32/v_cos_f32,
// whose argument is already in revolutions) — the two agree to a few 1e-7 absolute (tests/test_gpu_philox.py).
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define BF_HD __host__ __device__ __forceinline__
#define BF_D __device__ __forceinline__
#else
#define BF_HD static inline
#endif

#define BF_PHILOX_M0 0xD2511F53u
#define BF_PHILOX_M1 0xCD9E8D57u
#define BF_PHILOX_W0 0x9E3779B9u
#define BF_PHILOX_W1 0xBB67AE85u

struct bf_u32x4 {
    uint32_t x, y, z, w;
};

// Philox4x32 (Salmon et al., SC'11; Random123's philox4x32_R) with R = 7 rounds — the round count the authors
// certify as Crush-resistant (BigCrush passes from 7 rounds on; 10 is their default with extra margin).  The integer
// multiplies of the round function are the largest single cost of the VALU-bound sampling kernel: 7 instead of 10
// rounds takes 6 of 19 64-bit multiply-adds out of every block of 4 normals.  Round r uses key + r*W.
#ifndef BF_PHILOX_ROUNDS
#define BF_PHILOX_ROUNDS 7  // -DBF_PHILOX_ROUNDS=10 builds the Random123 default (tests/test_oracle_philox.py keeps that build pinned)
#endif
BF_HD bf_u32x4 bf_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < BF_PHILOX_ROUNDS; ++r) {
        const uint64_t p0 = (uint64_t)BF_PHILOX_M0 * c0;
        const uint64_t p1 = (uint64_t)BF_PHILOX_M1 * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += BF_PHILOX_W0;
        k1 += BF_PHILOX_W1;
    }
    bf_u32x4 o = {c0, c1, c2, c3};
    return o;
}

BF_HD float bf_u32_to_unit(uint32_t x) {
    // exact in both worlds: cvt u32->f32 (RNE) followed by one fused multiply-add
    return fmaf((float)x, 2.3283064365386963e-10f /* 2^-32 */, 1.1641532182693481e-10f /* 2^-33 */);
}

// Host twin: fp64 Box-Muller on the fp32 uniforms, rounded once to fp32.
static inline void bf_box_muller_host(uint32_t a, uint32_t b, float* z0, float* z1) {
    const double u1 = (double)bf_u32_to_unit(a);
    const double u2 = (double)bf_u32_to_unit(b);
    const double r = sqrt(-2.0 * log(u1));
    const double t = 6.283185307179586476925286766559 * u2;
    *z0 = (float)(r * cos(t));
    *z1 = (float)(r * sin(t));
}

static inline void bf_normal4_host(uint64_t group, uint32_t sample, uint32_t stream, uint64_t seed, float z[4]) {
    const bf_u32x4 x = bf_philox4x32((uint32_t)group, sample, stream, (uint32_t)(group >> 32), (uint32_t)seed,
                                        (uint32_t)(seed >> 32));
    bf_box_muller_host(x.x, x.y, &z[0], &z[1]);
    bf_box_muller_host(x.z, x.w, &z[2], &z[3]);
}

// ---- dropout contract -----------------------------------------------------------------------------------------------
// The reference trains with the wrapped model in .train() (/root/reference/examples/bert_glue.py:221): HuggingFace's
// dropout (p = 0.1) is a second, non-Philox noise source there.  Inside the fused kernels a keep / drop decision is a pure
// function of (seed, call, site, element), so a backward pass — or the recomputation of a checkpointed block — finds the
// forward's mask again without storing it:
//   group g = 8 elements (which 8 is the kernel's choice and part of its contract: 8 consecutive hidden features of a row
//             for the residual + LayerNorm path, the 8 probabilities of one P^T fragment for the attention path)
//   (x0..x3) = Philox4x32-7(counter = {lo32(g), call, 0x80000000 | site, hi32(g)}, key = {lo32(seed), hi32(seed)})
//   field i (0..7) = the 16-bit halves of x0, x1, x2, x3 in that order, low half first
//   element i is KEPT iff field_i >= thresh,  thresh = round(p * 65536)   (p exact to 1.5e-5), kept values scale by 1/(1-p')
//   with p' = thresh / 65536.
// Streams with the top bit set cannot collide with the weight streams 2 * layer_id + {0, 1}.  `call` identifies the
// forward (one number per bnn.Model forward), `site` the module the dropout belongs to.  The group index is GLOBAL over the
// step's Monte-Carlo samples: a kernel that sees the slabs of samples [s0, s0 + S_local) numbers its groups from
// first_group = s0 * (groups per sample) — masks are a function of the global sample, like epsilon.
#define BF_DROPOUT_STREAM 0x80000000u
BF_HD uint32_t bf_dropout_thresh(float p) {
    const float t = p * 65536.0f + 0.5f;
    return t <= 0.f ? 0u : (t >= 65535.f ? 65535u : (uint32_t)t);
}
BF_HD uint32_t bf_dropout_keep8(uint32_t g_lo, uint32_t g_hi, uint32_t call, uint32_t site, uint32_t k0, uint32_t k1,
                                uint32_t thresh) {
    const bf_u32x4 x = bf_philox4x32(g_lo, call, BF_DROPOUT_STREAM | site, g_hi, k0, k1);
    const uint32_t w[4] = {x.x, x.y, x.z, x.w};
    uint32_t keep = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        keep |= ((w[i] & 0xFFFFu) >= thresh ? 1u : 0u) << (2 * i);
        keep |= ((w[i] >> 16) >= thresh ? 1u : 0u) << (2 * i + 1);
    }
    return keep;
}

// what a kernel needs to know about a dropout it applies (thresh == 0: none)
struct bf_dropout_t {
    uint32_t k0, k1;     // seed
    uint32_t call, site;
    uint32_t thresh;     // bf_dropout_thresh(p)
    float inv_keep;      // 1 / (1 - thresh / 65536)
    // group index of the tensor's FIRST group in the step's GLOBAL numbering: an S-sharded rank passes (first global sample
    // of its shard) x (groups per sample), so that sample s draws the same masks whichever rank — or how many — run it
    uint32_t g0_lo, g0_hi;
    // device-resident part of `call` (NULL: none): the kernels use call + *d_call.  What makes a training step replayable from
    // a HIP graph: the graph bakes the host-side `call`, a one-element counter the captured step copies and increments moves
    // every replay on to fresh masks (the sample counter's scheme, bf_set_sample_counter)
    const uint32_t* d_call;
};

#if defined(__HIPCC__) || defined(__HIP__)
BF_D uint32_t bf_dropout_call(const bf_dropout_t& d) { return d.call + (d.d_call ? *d.d_call : 0u); }

// Device Box-Muller on the hardware transcendental units.
//   v_log_f32 is log2; v_sin_f32 / v_cos_f32 take their argument in revolutions (valid for |x| <= 256).
BF_D void bf_box_muller_dev(uint32_t a, uint32_t b, float& z0, float& z1) {
    const float u1 = bf_u32_to_unit(a);
    const float u2 = bf_u32_to_unit(b);
    // -2 ln(u1) = (-2 ln 2) * log2(u1)
    const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
    z0 = r * __builtin_amdgcn_cosf(u2);
    z1 = r * __builtin_amdgcn_sinf(u2);
}

// The same block function split at what does NOT depend on the Monte-Carlo sample index.  The counter is
// {lo32(group), sample, stream, hi32(group)}: the sample enters round 0 through c1 only, so in rounds 0-2 half of the
// products (M0 * lo32(group) in round 0, M1 * c2 in round 1, M0 * c0 in round 2) and the words they feed are the same
// for every sample of a group.  A kernel that draws S samples of the same elements computes them once
// (bf_philox_prepare) and pays 9 instead of 14 64-bit multiplies per further block (bf_philox_finish).  Bit-identical
// to bf_philox4x32 by construction (tests/test_gpu_philox.py compares the kernels that use it with the oracle).
struct bf_philox_inv {
    uint32_t c3a;   // round 0: lo(M0 * group_lo)                              -> c3 entering round 1
    uint32_t x1;    // round 1: lo(M1 * c2a)                 (c1 entering round 2; that round's key joins in bf_philox_finish)
    uint32_t h0k;   // round 2: hi(M0 * c0b) ^ (k1 + 2 W1)   (c0b = c0 entering round 2, sample-independent)
    uint32_t l0;    // round 2: lo(M0 * c0b)                                  -> c3 entering round 3
};

BF_D bf_philox_inv bf_philox_prepare(uint32_t group_lo, uint32_t group_hi, uint32_t stream, uint32_t k0, uint32_t k1) {
    static_assert(BF_PHILOX_ROUNDS >= 3, "the split assumes at least three rounds");
    const uint64_t p0 = (uint64_t)BF_PHILOX_M0 * group_lo;                    // round 0, counter word 0
    const uint32_t c2a = (uint32_t)(p0 >> 32) ^ group_hi ^ k1;                // c2 entering round 1
    const uint64_t p1s = (uint64_t)BF_PHILOX_M1 * stream;                     // round 0, counter word 2 (scalar)
    const uint32_t c1a = (uint32_t)p1s;                                       // c1 entering round 1
    const uint64_t q1 = (uint64_t)BF_PHILOX_M1 * c2a;                         // round 1
    const uint32_t c0b = (uint32_t)(q1 >> 32) ^ c1a ^ (k0 + BF_PHILOX_W0);    // c0 entering round 2
    const uint64_t r0 = (uint64_t)BF_PHILOX_M0 * c0b;                         // round 2
    bf_philox_inv v;
    v.c3a = (uint32_t)p0;
    v.x1 = (uint32_t)q1;
    v.h0k = (uint32_t)(r0 >> 32) ^ (k1 + 2u * BF_PHILOX_W1);
    v.l0 = (uint32_t)r0;
    return v;
}

// a ^ b ^ key in one instruction (gfx950's three-input bit operation; hipcc emits two v_xor for it).  `key` is the round
// key: wave-uniform by construction (derived from the seed), hence the scalar-register constraint.
BF_D uint32_t bf_xor3_key(uint32_t a, uint32_t b, uint32_t key) {
    uint32_t r;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(r) : "v"(a), "v"(b), "s"(key));
    return r;
}

BF_D bf_u32x4 bf_philox_finish(const bf_philox_inv& v, uint32_t sample, uint32_t stream, uint32_t k0, uint32_t k1) {
    // round 0, the sample-dependent word (uniform across the wave: scalar arithmetic)
    const uint64_t p1s = (uint64_t)BF_PHILOX_M1 * stream;
    const uint32_t c0a = (uint32_t)(p1s >> 32) ^ sample ^ k0;                 // c0 entering round 1
    // round 1: M0 * c0a is uniform too
    const uint64_t q0 = (uint64_t)BF_PHILOX_M0 * c0a;
    const uint32_t c2b = (uint32_t)(q0 >> 32) ^ v.c3a ^ (k1 + BF_PHILOX_W1);  // c2 entering round 2
    const uint32_t c3b = (uint32_t)q0;                                        // c3 entering round 2 (uniform)
    // round 2: one product left
    const uint64_t r1 = (uint64_t)BF_PHILOX_M1 * c2b;
    uint32_t c0 = bf_xor3_key((uint32_t)(r1 >> 32), v.x1, k0 + 2u * BF_PHILOX_W0);
    uint32_t c1 = (uint32_t)r1;
    uint32_t c2 = v.h0k ^ c3b;
    uint32_t c3 = v.l0;
    uint32_t ka = k0 + 3u * BF_PHILOX_W0, kb = k1 + 3u * BF_PHILOX_W1;
#pragma unroll
    for (int r = 3; r < BF_PHILOX_ROUNDS; ++r) {
        const uint64_t p0 = (uint64_t)BF_PHILOX_M0 * c0;
        const uint64_t p1 = (uint64_t)BF_PHILOX_M1 * c2;
        const uint32_t n0 = bf_xor3_key((uint32_t)(p1 >> 32), c1, ka);
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = bf_xor3_key((uint32_t)(p0 >> 32), c3, kb);
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        ka += BF_PHILOX_W0;
        kb += BF_PHILOX_W1;
    }
    bf_u32x4 o = {c0, c1, c2, c3};
    return o;
}

BF_D void bf_normal4_split_dev(const bf_philox_inv& v, uint32_t sample, uint32_t stream, uint32_t k0, uint32_t k1,
                               float z[4]) {
    const bf_u32x4 x = bf_philox_finish(v, sample, stream, k0, k1);
    bf_box_muller_dev(x.x, x.y, z[0], z[1]);
    bf_box_muller_dev(x.z, x.w, z[2], z[3]);
}

BF_D void bf_normal4_dev(uint32_t group_lo, uint32_t group_hi, uint32_t sample, uint32_t stream, uint32_t k0,
                         uint32_t k1, float z[4]) {
    const bf_u32x4 x = bf_philox4x32(group_lo, sample, stream, group_hi, k0, k1);
    bf_box_muller_dev(x.x, x.y, z[0], z[1]);
    bf_box_muller_dev(x.z, x.w, z[2], z[3]);
}
#endif

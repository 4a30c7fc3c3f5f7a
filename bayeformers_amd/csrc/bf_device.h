// bf_device.h — device-side helpers shared by the HIP translation units (fast transcendentals, wave reductions).
#pragma once
#include <hip/hip_runtime.h>

constexpr float kLogSqrt2Pi = 0.91893853320467274178f;
constexpr float kLn2 = 0.69314718055994531f;

// e^x with the argument reduction done in two pieces so the result is good to ~1e-7 relative for |x| <= 80.
__device__ __forceinline__ float exp_fast(float x) {
    const float y = x * 1.4426950408889634f;
    const float yh = __builtin_rintf(y);
    float r = fmaf(x, 1.4426950408889634f, -yh);   // exact residual of the rounded product
    r = fmaf(x, 1.9259629911266175e-8f, r);         // low part of log2(e)
    return __builtin_ldexpf(__builtin_amdgcn_exp2f(r), (int)yh);
}

// torch.nn.functional.softplus(beta=1, threshold=20): sigma = rho > 20 ? rho : log1p(exp(rho))
// (Gaussian.sigma, /root/reference/bayeformers/nn/parameters/gaussian.py:81-88)
__device__ __forceinline__ float softplus_fast(float rho) {
    const float t = exp_fast(fminf(rho, 21.0f));
    // small t: alternating series (t^7/7 < 2e-10 relative below 2^-5)
    const float ser = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, fmaf(t, -1.0f / 6.0f, 0.2f), -0.25f), 1.0f / 3.0f), -0.5f), 1.0f);
    // otherwise log(u) + (t - (u - 1)) / u with u = fl(1 + t): the second term restores the bits lost in u
    const float u = 1.0f + t;
    const float big = fmaf(kLn2, __builtin_amdgcn_logf(u), (t - (u - 1.0f)) * __builtin_amdgcn_rcpf(u));
    const float sp = t < 0.03125f ? ser : big;
    return rho > 20.0f ? rho : sp;
}

__device__ __forceinline__ float log_fast(float x) { return kLn2 * __builtin_amdgcn_logf(x); }

// wave64 sum on the DPP network: row_shr 1,2,4,8 then row_bcast 15 / 31; the total lands in lane 63 and is
// broadcast to every lane.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    const int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, true);
    return v + __builtin_bit_cast(float, t);
}
// the four DPP steps inside a row of 16 lanes: lanes 15, 31, 47 and 63 end up with their row's sum
__device__ __forceinline__ float row_sum(float v) {
    v = dpp_add<0x111, 0xf>(v);  // row_shr:1
    v = dpp_add<0x112, 0xf>(v);  // row_shr:2
    v = dpp_add<0x114, 0xf>(v);  // row_shr:4
    return dpp_add<0x118, 0xf>(v);  // row_shr:8
}
// sum over each 32-lane half of the wave, returned to every lane of that half
__device__ __forceinline__ float half_sum(float v, int lane) {
    v = row_sum(v);
    v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3: lanes 31 / 63 hold their half's sum
    const float lo = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 31));
    const float hi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
    return lane < 32 ? lo : hi;
}
__device__ __forceinline__ float wave_sum(float v) {
    v = dpp_add<0x111, 0xf>(v);  // row_shr:1
    v = dpp_add<0x112, 0xf>(v);  // row_shr:2
    v = dpp_add<0x114, 0xf>(v);  // row_shr:4
    v = dpp_add<0x118, 0xf>(v);  // row_shr:8   -> lane 15 of each row holds the row sum
    v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave sum
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

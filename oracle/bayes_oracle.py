"""bayes_oracle — TEST INFRASTRUCTURE ONLY (oracle).  Never imported by the product path.

CPU restatement (torch-CPU fp32, plus fp64 closed forms) of the reference's hot path, one function per reference
symbol, each citing the file:line it follows under /root/reference/.  The op ORDER of the fp32 functions is the
reference's, so they reproduce its rounding; the *_f64 functions are the analytic values the HIP kernels are
held to.

Pinning: the reference has no tests.  This oracle is pinned against outputs of the real
reference, imported in the build container with this module's Philox epsilon injected at its only RNG
touch-point (gaussian.py:100) — see tests/golden/make_golden.py and tests/test_oracle_golden.py — and its
Philox against the Random123 known-answer vectors (tests/test_oracle_philox.py).
"""
import ctypes
import math
import os

import numpy as np
import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

LOG_SQRT_2PI = float(np.log(np.sqrt(2 * np.pi)))


# ------------------------------------------------------------------------------------------------ epsilon
def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "liboracle.so")
        if not os.path.exists(path):
            from oracle import build as _b  # compiled on demand in test processes

            _b.build()
        lib = ctypes.CDLL(path)
        lib.oracle_normals.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32,
                                       ctypes.c_uint32, ctypes.c_uint64]
        lib.oracle_normals.restype = None
        lib.oracle_philox4x32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        lib.oracle_philox4x32.restype = None
        _LIB = lib
    return _LIB


PHILOX_ROUNDS = 7  # the epsilon contract (csrc/bf_philox.h BF_PHILOX_ROUNDS, oracle/philox_oracle.c)


def philox4x32(ctr, key, rounds=PHILOX_ROUNDS):
    """Raw Philox4x32-R block (C implementation, oracle/philox_oracle.c)."""
    c = np.asarray(ctr, dtype=np.uint32).copy()
    k = np.asarray(key, dtype=np.uint32).copy()
    out = np.zeros(4, dtype=np.uint32)
    _lib().oracle_philox4x32(c.ctypes.data, k.ctypes.data, int(rounds), out.ctypes.data)
    return out


def philox4x32_numpy(ctr, key, rounds=PHILOX_ROUNDS):
    """Vectorised numpy Philox4x32-R, written independently of the C one.  ctr: [...,4] uint32, key: [2]."""
    c = np.asarray(ctr, dtype=np.uint64).copy()
    k0, k1 = np.uint64(key[0]), np.uint64(key[1])
    m0, m1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
    mask = np.uint64(0xFFFFFFFF)
    sh = np.uint64(32)
    for r in range(rounds):
        p0 = m0 * c[..., 0]
        p1 = m1 * c[..., 2]
        n0 = (p1 >> sh) ^ c[..., 1] ^ k0
        n1 = p1 & mask
        n2 = (p0 >> sh) ^ c[..., 3] ^ k1
        n3 = p0 & mask
        c = np.stack([n0, n1, n2, n3], axis=-1)
        k0 = (k0 + np.uint64(0x9E3779B9)) & mask
        k1 = (k1 + np.uint64(0xBB67AE85)) & mask
    return c.astype(np.uint32)


def dropout_keep(first_group, n_groups, p, seed, call, site):
    """Keep flags (uint8 [n_groups, 8], 1 = kept) of dropout groups first_group .. — an independent restatement of the
    dropout contract of bayeformers_amd/csrc/bf_philox.h: Philox4x32-7 with counter {lo32(g), call, 0x80000000 | site,
    hi32(g)}, the eight 16-bit halves of its four words (low half first) compared with round(p * 65536).

    Stands in for torch.nn.Dropout in the wrapped HuggingFace modules the reference trains in .train() mode
    (/root/reference/examples/bert_glue.py:221): like epsilon, the mask is the build's own contract — a test applies THIS
    mask with plain torch ops and compares with the kernels."""
    g = np.uint64(first_group) + np.arange(int(n_groups), dtype=np.uint64)
    ctr = np.stack([g & np.uint64(0xFFFFFFFF), np.full_like(g, np.uint64(call & 0xFFFFFFFF)),
                    np.full_like(g, np.uint64(0x80000000 | (site & 0x7FFFFFFF))), g >> np.uint64(32)], axis=-1)
    x = philox4x32_numpy(ctr, [seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF]).astype(np.uint32)
    fields = np.stack([f for i in range(4) for f in (x[:, i] & np.uint32(0xFFFF), x[:, i] >> np.uint32(16))], axis=-1)
    t = np.float32(p) * np.float32(65536.0) + np.float32(0.5)
    thresh = 0 if t <= 0 else (65535 if t >= 65535 else int(t))
    return (fields >= np.uint32(thresh)).astype(np.uint8)


def dropout_keep_scale(p):
    t = np.float32(p) * np.float32(65536.0) + np.float32(0.5)
    thresh = 0 if t <= 0 else (65535 if t >= 65535 else int(t))
    return 1.0 / (1.0 - thresh / 65536.0)


def attention_keep_mask(B, H, T, p, seed, call, site):
    """[B, H, T(query), T(key)] keep mask of bf_attention_fwd_dropout: the group of probability (b, h, q, key) is
    (((b*H + h)*T + q) * (T/32) + (key // 128)*4 + c) * 4 + lg with key % 128 = (2c + e)*16 + 4 lg + j, field e*4 + j."""
    keep = dropout_keep(0, B * H * T * (T // 8), p, seed, call, site).reshape(B, H, T, T // 128, 4, 4, 8)  # [.., tile, c, lg, field]
    out = np.empty((B, H, T, T), dtype=np.uint8)
    for tile in range(T // 128):
        for c in range(4):
            for lg in range(4):
                for e in range(2):
                    k0 = tile * 128 + (2 * c + e) * 16 + 4 * lg
                    out[..., k0:k0 + 4] = keep[:, :, :, tile, c, lg, e * 4:e * 4 + 4]
    return out


def normals(n, seed, sample, stream, offset=0):
    """eps for elements [offset, offset+n) of Philox stream `stream`, MC sample `sample` (fp32 numpy array).

    Stands in for Normal(0,1).sample(size), /root/reference/bayeformers/nn/parameters/gaussian.py:100."""
    out = np.empty(int(n), dtype=np.float32)
    _lib().oracle_normals(out.ctypes.data, int(n), int(seed) & (2**64 - 1), int(sample) & 0xFFFFFFFF,
                          int(stream) & 0xFFFFFFFF, int(offset))
    return out


def normals_numpy(n, seed, sample, stream, offset=0):
    """Same contract as normals(), pure numpy (cross-check of the C code; use for small n)."""
    e = np.arange(offset, offset + n, dtype=np.uint64)
    g = e >> np.uint64(2)
    ctr = np.stack([g & np.uint64(0xFFFFFFFF), np.full_like(g, sample & 0xFFFFFFFF),
                    np.full_like(g, stream & 0xFFFFFFFF), g >> np.uint64(32)], axis=-1)
    x = philox4x32_numpy(ctr, [seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF])
    j = (e & np.uint64(3)).astype(np.int64)
    pair = j >> 1
    a = np.take_along_axis(x, (2 * pair)[:, None], axis=1)[:, 0]
    b = np.take_along_axis(x, (2 * pair + 1)[:, None], axis=1)[:, 0]
    # u = fma(float(x), 2^-32, 2^-33) in fp32: float(x) is exact-rounded, the scaled product is exact (power of two),
    # so a single fp32 rounding of the fp64 sum reproduces the fused op.
    u1 = (a.astype(np.float32).astype(np.float64) * 2.0**-32 + 2.0**-33).astype(np.float32).astype(np.float64)
    u2 = (b.astype(np.float32).astype(np.float64) * 2.0**-32 + 2.0**-33).astype(np.float32).astype(np.float64)
    r = np.sqrt(-2.0 * np.log(u1))
    t = 2.0 * np.pi * u2
    return np.where((j & 1) == 0, r * np.cos(t), r * np.sin(t)).astype(np.float32)


def eps_tensor(shape, seed, sample, layer_id, tensor_id):
    """eps of one parameter tensor as a torch fp32 tensor; stream = 2*layer_id + tensor_id (0 weight, 1 bias)."""
    n = int(np.prod(shape))
    return torch.from_numpy(normals(n, seed, sample, 2 * layer_id + tensor_id)).reshape(tuple(shape))


# ------------------------------------------------------------------------------------------------ a1..a4
def sigma(rho):
    """Gaussian.sigma — /root/reference/bayeformers/nn/parameters/gaussian.py:81-88."""
    return F.softplus(rho)


def gaussian_sample(mu, rho, eps):
    """Gaussian.sample with eps given — gaussian.py:90-101: mu + eps * sigma."""
    return mu + eps * sigma(rho)


def gaussian_log_prob(x, mu, rho):
    """Gaussian.log_prob — gaussian.py:103-116, same expression order, fp32."""
    s = sigma(rho)
    return (-LOG_SQRT_2PI - torch.log(s) - ((x - mu) ** 2) / (2 * s ** 2)).sum()


def mixture_log_prob(x, pi, sigma1, sigma2):
    """ScaledGaussianMixture.log_prob — gaussian.py:160-171 (fp32; underflows to -inf for |x| >= 14.3 at sigma1=1)."""
    zero = torch.tensor(0.0)
    g1 = torch.distributions.Normal(zero, torch.tensor(float(sigma1)).float())
    g2 = torch.distributions.Normal(zero, torch.tensor(float(sigma2)).float())
    pi_t = torch.tensor(float(pi)).float()
    p1 = torch.exp(g1.log_prob(x))
    p2 = torch.exp(g2.log_prob(x))
    return torch.log(pi_t * p1 + (1.0 - pi_t) * p2).sum()


def gaussian_log_prob_terms_f64(eps, mu, rho, x=None):
    """Per-element closed form in fp64.  As the posterior of its own sample (x is None) the quadratic term is
    eps^2/2."""
    s = F.softplus(rho.double())
    if x is None:
        q = 0.5 * eps.double() ** 2
    else:
        q = (x.double() - mu.double()) ** 2 / (2 * s ** 2)
    return -LOG_SQRT_2PI - torch.log(s) - q


def gaussian_log_prob_f64(eps, mu, rho, x=None):
    return float(gaussian_log_prob_terms_f64(eps, mu, rho, x).sum())


def mixture_log_prob_terms_f64(x, pi, sigma1, sigma2):
    """Per-element log(pi N(x;0,s1) + (1-pi) N(x;0,s2)) in fp64, log-sum-exp form (finite for all x)."""
    x = x.double()
    pi = float(np.float32(pi))
    s1 = float(np.float32(sigma1))
    s2 = float(np.float32(sigma2))
    t1 = -0.5 * (x / s1) ** 2 - math.log(s1) - LOG_SQRT_2PI + (math.log(pi) if pi > 0 else -math.inf)
    t2 = -0.5 * (x / s2) ** 2 - math.log(s2) - LOG_SQRT_2PI + (math.log1p(-pi) if pi < 1 else -math.inf)
    return torch.logaddexp(t1, t2)


def mixture_log_prob_f64(x, pi, sigma1, sigma2):
    return float(mixture_log_prob_terms_f64(x, pi, sigma1, sigma2).sum())


def linear_logprob_magnitudes(mu_w, rho_w, mu_b, rho_b, eps_w, eps_b, prior_w, prior_b=None):
    """(sum |log-prior terms|, sum |log-q terms|): the scale a summation tolerance is relative to — the signed
    sums can cancel to nearly zero (e.g. a narrow custom mixture) while every term is O(1)."""
    if prior_b is None:
        prior_b = prior_w

    def lp_terms(prior, w):
        if prior is None or w is None:
            return torch.zeros(1, dtype=torch.float64)
        if prior[0] == "mixture":
            return mixture_log_prob_terms_f64(w, *prior[1:])
        return gaussian_log_prob_terms_f64(None, prior[1], prior[2], x=w)

    W = mu_w.double() + eps_w.double() * F.softplus(rho_w.double())
    ap = float(lp_terms(prior_w, W).abs().sum())
    aq = float(gaussian_log_prob_terms_f64(eps_w, mu_w, rho_w).abs().sum())
    if mu_b is not None:
        b = mu_b.double() + eps_b.double() * F.softplus(rho_b.double())
        ap += float(lp_terms(prior_b, b).abs().sum())
        aq += float(gaussian_log_prob_terms_f64(eps_b, mu_b, rho_b).abs().sum())
    return ap, aq


# ------------------------------------------------------------------------------------------------ a6
def linear_forward(x, mu_w, rho_w, mu_b, rho_b, eps_w, eps_b, prior_w, prior_b=None):
    """Linear.forward — /root/reference/bayeformers/nn/layers/linear.py:83-104 with eps given.

    prior_* is ("mixture", pi, s1, s2) | ("gaussian", mu_p, rho_p) | None (NoneParameter, base.py:55-69).
    Returns (y, log_prior, log_variational_posterior) as fp32 tensors."""
    if prior_b is None:
        prior_b = prior_w

    def lp(prior, w):
        if prior is None or w is None:
            return torch.tensor(0.0)
        if prior[0] == "mixture":
            return mixture_log_prob(w, *prior[1:])
        return gaussian_log_prob(w, prior[1], prior[2])

    W = gaussian_sample(mu_w, rho_w, eps_w)
    b = gaussian_sample(mu_b, rho_b, eps_b) if mu_b is not None else None
    log_prior = lp(prior_w, W) + (lp(prior_b, b) if b is not None else 0.0)
    lvp = gaussian_log_prob(W, mu_w, rho_w) + (gaussian_log_prob(b, mu_b, rho_b) if b is not None else 0.0)
    return F.linear(x, W, b), log_prior, lvp


def linear_logprobs_f64(mu_w, rho_w, mu_b, rho_b, eps_w, eps_b, prior_w, prior_b=None, rounded_w=False):
    """fp64 closed-form (log_prior, log_variational_posterior) of one layer for one sample.

    rounded_w=False: the analytic value, quadratic term eps^2/2 (what the HIP kernel computes).
    rounded_w=True : the reference's route, (W - mu)^2 / (2 sigma^2) with W = fp32(mu + eps*sigma) — differs from
    the analytic value only where sigma*eps is below the fp32 resolution of mu (cancellation)."""
    if prior_b is None:
        prior_b = prior_w
    if rounded_w:
        W32 = gaussian_sample(mu_w, rho_w, eps_w)
        lq = gaussian_log_prob_f64(None, mu_w, rho_w, x=W32)
        lpr = (mixture_log_prob_f64(W32, *prior_w[1:]) if prior_w[0] == "mixture"
               else gaussian_log_prob_f64(None, prior_w[1], prior_w[2], x=W32)) if prior_w is not None else 0.0
        if mu_b is not None:
            b32 = gaussian_sample(mu_b, rho_b, eps_b)
            lq += gaussian_log_prob_f64(None, mu_b, rho_b, x=b32)
            if prior_b is not None:
                lpr += (mixture_log_prob_f64(b32, *prior_b[1:]) if prior_b[0] == "mixture"
                        else gaussian_log_prob_f64(None, prior_b[1], prior_b[2], x=b32))
        return lpr, lq

    def lp(prior, w):
        if prior is None or w is None:
            return 0.0
        if prior[0] == "mixture":
            return mixture_log_prob_f64(w, *prior[1:])
        return gaussian_log_prob_f64(None, prior[1], prior[2], x=w)

    W = mu_w.double() + eps_w.double() * F.softplus(rho_w.double())
    lq = gaussian_log_prob_f64(eps_w, mu_w, rho_w)
    lpr = lp(prior_w, W)
    if mu_b is not None:
        b = mu_b.double() + eps_b.double() * F.softplus(rho_b.double())
        lq += gaussian_log_prob_f64(eps_b, mu_b, rho_b)
        lpr += lp(prior_b, b)
    return lpr, lq


# ------------------------------------------------------------------------------------------------ a11
def moped_rho(w, delta):
    """MOPED rho — layers/linear.py:141-144: log(exp(delta*|w|) - 1), -inf -> 0.0 (fp32)."""
    rho = torch.log(torch.exp(delta * torch.abs(w)) - 1.0)
    rho[rho == float("-inf")] = 0.0
    return rho


# ------------------------------------------------------------------------------------------------ a10
def elbo(mean_log_prior, mean_lvp, nll, n_batches):
    """loss = (lvp - log_prior) / n_batches + nll — examples/bert_glue.py:235, examples/mlp_mnist.py:107."""
    return (mean_lvp - mean_log_prior) / n_batches + nll


def cpu_reference_step(x, mu_w, rho_w, mu_b, rho_b, S, prior_w):
    """The reference's op sequence for S serial samples of ONE layer with torch's own RNG (timing baseline only):
    normal -> softplus -> mul/add -> log-prob reductions -> F.linear  (linear.py:97-104, gaussian.py:100-116)."""
    outs = []
    lps = []
    lqs = []
    for _ in range(S):
        eps_w = torch.randn(mu_w.shape)
        eps_b = torch.randn(mu_b.shape) if mu_b is not None else None
        y, lp, lq = linear_forward(x, mu_w, rho_w, mu_b, rho_b, eps_w, eps_b, prior_w)
        outs.append(y)
        lps.append(lp)
        lqs.append(lq)
    return torch.stack(outs), torch.stack(lps), torch.stack(lqs)

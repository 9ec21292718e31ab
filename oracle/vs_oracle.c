/*
 * vs_oracle.c -- CPU restatement of the arithmetic primitives on VISinger's variational-inference hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/, the smoke check in
 * __graft_entry__.py and the cpu_baseline leg of bench.py may load this library, and only as the checker /
 * the reported CPU baseline.  The product path (visinger_amd/) never falls back to it.
 *
 * Parity: PINNED.  Every primitive here is checked (tests/test_oracle_golden.py) against golden vectors that
 * tests/golden/make_golden.py produced by importing the reference's own PyTorch modules.
 *
 * Built twice from this one file (oracle/Makefile):
 *   -DREAL=double -> libvs_oracle_f64.so  (referee: fp64 data and accumulation)
 *   -DREAL=float  -> libvs_oracle_f32.so  (the "port" CPU baseline: same arithmetic type as the reference)
 *
 * Reference sites restated (paths under /root/reference):
 *   weight-norm        torch.nn.utils.weight_norm as applied at modules/visinger/encoder.py:147,154,164 and
 *                      modules/visinger/decoder.py:24,72-87  (w = g * v / ||v||_2 over all dims but dim 0)
 *   conv1d             nn.Conv1d sites: encoder.py:88,90,152,163; flow.py:60,62; decoder.py:19,34,72-87;
 *                      rel_transformer.py:120-128,332-333
 *   conv_transpose1d   decoder.py:24-26,47
 *   layernorm_c        rel_transformer.py:33-42   (biased variance over channels, eps inside rsqrt)
 *   rel_attention      rel_transformer.py:148-179 (+ helpers 181-243)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

#ifndef REAL
#define REAL double
#endif
typedef REAL real;

#define API __attribute__((visibility("default")))

API int orc_real_bytes(void) { return (int)sizeof(real); }

#ifdef _OPENMP
#include <omp.h>
/* threads of the parallel loops below (the CPU-baseline leg sets it to the cores the process may actually use) */
API int orc_set_threads(int n) { if (n > 0) omp_set_num_threads(n); return omp_get_max_threads(); }
#else
API int orc_set_threads(int n) { (void)n; return 1; }
#endif

/* w[r, :] = g[r] * v[r, :] / ||v[r, :]||_2      (rows = dim 0 of the weight, cols = product of the rest) */
API void orc_weightnorm(const real *v, const real *g, real *w, int64_t rows, int64_t cols) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < rows; ++r) {
        double ss = 0.0;
        for (int64_t c = 0; c < cols; ++c) ss += (double)v[r * cols + c] * (double)v[r * cols + c];
        real s = (real)((double)g[r] / sqrt(ss));
        for (int64_t c = 0; c < cols; ++c) w[r * cols + c] = v[r * cols + c] * s;
    }
}

/* y[b, co, t] = bias[co] + sum_{ci, k} w[co, ci, k] * x[b, ci, t + k*dil - pad]   (zero padding, stride 1)
 * x: [B, Cin, T]   w: [Cout, Cin, K]   y: [B, Cout, Tout],  Tout = T + 2*pad - dil*(K-1) */
#define ORC_TCHUNK 2048   /* time-axis chunk: the parallel loop runs over (b, co, chunk) so that B=1 still fills the cores */

API void orc_conv1d(const real *x, const real *w, const real *bias, real *y, int64_t B, int64_t Cin, int64_t Cout,
                    int64_t T, int64_t K, int64_t dil, int64_t pad) {
    const int64_t Tout = T + 2 * pad - dil * (K - 1);
    const int64_t nchunk = (Tout + ORC_TCHUNK - 1) / ORC_TCHUNK;
#pragma omp parallel for collapse(3) schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        for (int64_t co = 0; co < Cout; ++co) {
            for (int64_t ch = 0; ch < nchunk; ++ch) {
                const int64_t c0 = ch * ORC_TCHUNK, c1 = c0 + ORC_TCHUNK < Tout ? c0 + ORC_TCHUNK : Tout;
                real *yr = y + (b * Cout + co) * Tout;
                const real b0 = bias ? bias[co] : (real)0;
                for (int64_t t = c0; t < c1; ++t) yr[t] = b0;
                for (int64_t ci = 0; ci < Cin; ++ci) {
                    const real *xr = x + (b * Cin + ci) * T;
                    for (int64_t k = 0; k < K; ++k) {
                        const real wv = w[(co * Cin + ci) * K + k];
                        const int64_t off = k * dil - pad;          /* x index = t + off */
                        int64_t t0 = off < 0 ? -off : 0;
                        int64_t t1 = T - off < Tout ? T - off : Tout;
                        if (t0 < c0) t0 = c0;
                        if (t1 > c1) t1 = c1;
                        for (int64_t t = t0; t < t1; ++t) yr[t] += wv * xr[t + off];
                    }
                }
            }
        }
    }
}

/* y[b, co, n] = bias[co] + sum_{ci, m, k : n = m*stride - pad + k} x[b, ci, m] * w[ci, co, k]
 * x: [B, Cin, T]   w: [Cin, Cout, K]   y: [B, Cout, Tout],  Tout = (T-1)*stride - 2*pad + K */
API void orc_conv_transpose1d(const real *x, const real *w, const real *bias, real *y, int64_t B, int64_t Cin,
                              int64_t Cout, int64_t T, int64_t K, int64_t stride, int64_t pad) {
    const int64_t Tout = (T - 1) * stride - 2 * pad + K;
    const int64_t nchunk = (Tout + ORC_TCHUNK - 1) / ORC_TCHUNK;
#pragma omp parallel for collapse(3) schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        for (int64_t co = 0; co < Cout; ++co) {
            for (int64_t ch = 0; ch < nchunk; ++ch) {
                const int64_t c0 = ch * ORC_TCHUNK, c1 = c0 + ORC_TCHUNK < Tout ? c0 + ORC_TCHUNK : Tout;
                real *yr = y + (b * Cout + co) * Tout;
                const real b0 = bias ? bias[co] : (real)0;
                for (int64_t n = c0; n < c1; ++n) yr[n] = b0;
                for (int64_t ci = 0; ci < Cin; ++ci) {
                    const real *xr = x + (b * Cin + ci) * T;
                    for (int64_t k = 0; k < K; ++k) {
                        const real wv = w[(ci * Cout + co) * K + k];
                        /* n = m*stride - pad + k in [c0, c1)  <=>  m in [ceil((c0+pad-k)/stride), ceil((c1+pad-k)/stride)) */
                        int64_t lo = c0 + pad - k, hi = c1 + pad - k;
                        int64_t m0 = lo <= 0 ? 0 : (lo + stride - 1) / stride;
                        int64_t m1 = hi <= 0 ? 0 : (hi + stride - 1) / stride;
                        if (m1 > T) m1 = T;
                        for (int64_t m = m0; m < m1; ++m) yr[m * stride - pad + k] += wv * xr[m];
                    }
                }
            }
        }
    }
}

/* LayerNorm over the channel dim of [B, C, T]: biased variance, eps inside rsqrt. */
API void orc_layernorm_c(const real *x, const real *gamma, const real *beta, real *y, int64_t B, int64_t C,
                         int64_t T, double eps) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        for (int64_t t = 0; t < T; ++t) {
            const real *xb = x + b * C * T + t;
            real mean = 0;
            for (int64_t c = 0; c < C; ++c) mean += xb[c * T];
            mean /= (real)C;
            real var = 0;
            for (int64_t c = 0; c < C; ++c) { real d = xb[c * T] - mean; var += d * d; }
            var /= (real)C;
            const real rs = (real)(1.0 / sqrt((double)var + eps));
            for (int64_t c = 0; c < C; ++c) y[b * C * T + c * T + t] = (xb[c * T] - mean) * rs * gamma[c] + beta[c];
        }
    }
}

/* Windowed relative-position self-attention core.
 *   q, k, v : [B, nh*dk, T]  (channel-major, as the 1x1 convs produce them)
 *   rel_k, rel_v : [nh_rel, 2*ws+1, dk]  (nh_rel = 1 when heads share) or NULL when ws < 0
 *   mask : [B, T] (1 = valid) or NULL; attn mask[i, j] = mask[i] * mask[j]; masked scores are SET to -1e4
 *   out  : [B, nh*dk, T];   p_out : [B, nh, T, T] or NULL
 * scores[i, j] = (q_i . k_j + [|j-i| <= ws] q_i . rel_k[j-i+ws]) / sqrt(dk)
 * out_i = sum_j p[i, j] v_j + sum_{|d| <= ws, 0 <= i+d < T} p[i, i+d] rel_v[d+ws]
 * (the reference's pad/reshape "skew" is exactly this index map; entries outside the window multiply zeros) */
API void orc_rel_attention(const real *q, const real *k, const real *v, const real *rel_k, const real *rel_v,
                           const real *mask, real *out, real *p_out, int64_t B, int64_t nh, int64_t dk, int64_t T,
                           int64_t ws, int64_t nh_rel) {
    const real scale = (real)(1.0 / sqrt((double)dk));
#pragma omp parallel for collapse(2) schedule(dynamic)
    for (int64_t b = 0; b < B; ++b) {
        for (int64_t h = 0; h < nh; ++h) {
            const real *qh = q + (b * nh + h) * dk * T;
            const real *kh = k + (b * nh + h) * dk * T;
            const real *vh = v + (b * nh + h) * dk * T;
            const real *rk = (ws >= 0 && rel_k) ? rel_k + (nh_rel == 1 ? 0 : h) * (2 * ws + 1) * dk : NULL;
            const real *rv = (ws >= 0 && rel_v) ? rel_v + (nh_rel == 1 ? 0 : h) * (2 * ws + 1) * dk : NULL;
            real *p = (real *)malloc(sizeof(real) * (size_t)T);
            for (int64_t i = 0; i < T; ++i) {
                for (int64_t j = 0; j < T; ++j) {
                    real s = 0;
                    for (int64_t d = 0; d < dk; ++d) s += qh[d * T + i] * kh[d * T + j];
                    s *= scale;
                    const int64_t r = j - i;
                    if (rk && r >= -ws && r <= ws) {
                        real sr = 0;
                        for (int64_t d = 0; d < dk; ++d) sr += qh[d * T + i] * rk[(r + ws) * dk + d];
                        s += sr * scale;
                    }
                    if (mask && (mask[b * T + i] * mask[b * T + j]) == (real)0) s = (real)-1e4;
                    p[j] = s;
                }
                real mx = p[0];
                for (int64_t j = 1; j < T; ++j) mx = p[j] > mx ? p[j] : mx;
                real den = 0;
                for (int64_t j = 0; j < T; ++j) { p[j] = (real)exp((double)(p[j] - mx)); den += p[j]; }
                for (int64_t j = 0; j < T; ++j) p[j] /= den;
                if (p_out) memcpy(p_out + (((b * nh + h) * T + i) * T), p, sizeof(real) * (size_t)T);
                for (int64_t d = 0; d < dk; ++d) {
                    real o = 0;
                    for (int64_t j = 0; j < T; ++j) o += p[j] * vh[d * T + j];
                    if (rv) {
                        for (int64_t r = -ws; r <= ws; ++r) {
                            const int64_t j = i + r;
                            if (j >= 0 && j < T) o += p[j] * rv[(r + ws) * dk + d];
                        }
                    }
                    out[(b * nh + h) * dk * T + d * T + i] = o;
                }
            }
            free(p);
        }
    }
}

/* Strided / grouped conv1d for the discriminators (modules/discriminator.py:20-27,55-63):
 * y[b, co, t] = bias[co] + sum_{ci in group(co), k} w[co, ci_local, k] * x[b, ci, t*stride + k - pad]
 * x: [B, Cin, T]   w: [Cout, Cin/groups, K]   y: [B, Cout, Tout],  Tout = (T + 2*pad - K) / stride + 1 */
API void orc_conv1d_sg(const real *x, const real *w, const real *bias, real *y, int64_t B, int64_t Cin, int64_t Cout,
                       int64_t T, int64_t K, int64_t stride, int64_t pad, int64_t groups) {
    const int64_t Tout = (T + 2 * pad - K) / stride + 1;
    const int64_t cig = Cin / groups, cog = Cout / groups;
#pragma omp parallel for collapse(2) schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        for (int64_t co = 0; co < Cout; ++co) {
            const int64_t g = co / cog;
            real *yr = y + (b * Cout + co) * Tout;
            for (int64_t t = 0; t < Tout; ++t) {
                real acc = bias ? bias[co] : (real)0;
                for (int64_t cl = 0; cl < cig; ++cl) {
                    const real *xr = x + (b * Cin + g * cig + cl) * T;
                    const real *wr = w + (co * cig + cl) * K;
                    for (int64_t k = 0; k < K; ++k) {
                        const int64_t n = t * stride + k - pad;
                        if (n >= 0 && n < T) acc += wr[k] * xr[n];
                    }
                }
                yr[t] = acc;
            }
        }
    }
}

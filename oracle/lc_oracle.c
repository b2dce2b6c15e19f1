/*
 * lc_oracle.c — CPU oracle (plain C + OpenMP).  TEST INFRASTRUCTURE ONLY — see lc_oracle.h.
 * Every function cites the reference file:line whose definition / arithmetic order it follows.
 */
#include "lc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---------------------------------------------------------------------------------------------- */
/* IEEE binary16 conversions (the reference relies on CUDA's __half2float / __float2half_rn).      */

float lc_h2f(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  const uint32_t exp = (h >> 10) & 0x1fu;
  const uint32_t man = h & 0x3ffu;
  uint32_t bits;
  if (exp == 0) {
    if (man == 0) {
      bits = sign;
    } else { /* subnormal: value = man * 2^-24 */
      float f = (float)man * 5.9604644775390625e-08f;
      memcpy(&bits, &f, 4);
      bits |= sign;
    }
  } else if (exp == 31) {
    bits = sign | 0x7f800000u | (man << 13);
  } else {
    bits = sign | ((exp + 112u) << 23) | (man << 13);
  }
  float out;
  memcpy(&out, &bits, 4);
  return out;
}

uint16_t lc_d2h(double d) {
  uint16_t sign = 0;
  if (d != d) return 0x7e00u;
  if (signbit(d)) {
    sign = 0x8000u;
    d = -d;
  }
  if (d == 0.0) return sign;
  if (isinf(d)) return (uint16_t)(sign | 0x7c00u);
  int e;
  (void)frexp(d, &e); /* d = m * 2^e, m in [0.5,1)  ->  floor(log2 d) = e-1 */
  int le = e - 1;
  if (le < -14) le = -14;               /* subnormal range shares the 2^-24 quantum */
  const double q = ldexp(d, 10 - le);   /* in [1024, 2048) for normals, [0,1024) for subnormals */
  double r = nearbyint(q);              /* round-to-nearest-even in the default rounding mode */
  if (r >= 2048.0) {
    r *= 0.5;
    le += 1;
  }
  if (le > 15) return (uint16_t)(sign | 0x7c00u); /* overflow -> inf */
  const uint32_t ri = (uint32_t)r;
  if (ri < 1024u) return (uint16_t)(sign | ri);    /* subnormal (or zero) */
  return (uint16_t)(sign | ((uint32_t)(le + 15) << 10) | (ri - 1024u));
}

uint16_t lc_f2h(float f) { return lc_d2h((double)f); } /* float -> double is exact */

static inline float rh(float x) { return lc_h2f(lc_f2h(x)); } /* round a float to fp16 precision */

int lc_oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ---------------------------------------------------------------------------------------------- */
/* HGEMM                                                                                           */

static int g_decode_bf16 = 0; /* set only inside lc_oracle_attn_exact_f32_bf16 (single caller at a time) */

static inline float bf16_to_f32(uint16_t b) {
  uint32_t u = (uint32_t)b << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

static float* to_f32(const uint16_t* x, size_t n) {
  float* out = (float*)malloc(n * sizeof(float));
  if (!out) return NULL;
  const int bf = g_decode_bf16;
#pragma omp parallel for schedule(static)
  for (long long i = 0; i < (long long)n; ++i) out[i] = bf ? bf16_to_f32(x[i]) : lc_h2f(x[i]);
  return out;
}

/* Bt[n][k] (k contiguous) from either storage, so the inner loop is a dot of two contiguous rows. */
static float* b_as_nk(const uint16_t* B, int N, int K, int layout) {
  float* bt = (float*)malloc((size_t)N * K * sizeof(float));
  if (!bt) return NULL;
  if (layout == 1) { /* TN: stored [N][K] already (kernels/hgemm/tools/utils.py:152-156) */
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < (long long)N * K; ++i) bt[i] = lc_h2f(B[i]);
  } else {
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; ++n)
      for (int k = 0; k < K; ++k) bt[(size_t)n * K + k] = lc_h2f(B[(size_t)k * N + n]);
  }
  return bt;
}

/* mode 0: fp16 out (one rounding), mode 1: fp32 out, mode 2: reference numerics (fp16 accumulator,
 * rounded after each K16 step as mma.sync.m16n8k16.f16.f16.f16.f16 does, hgemm_mma_stage.cu:110-116). */
static void hgemm_impl(const uint16_t* A, const uint16_t* B, void* C, int M, int N, int K, int layout,
                       int mode) {
  float* a = to_f32(A, (size_t)M * K);
  float* bt = b_as_nk(B, N, K, layout);
  if (!a || !bt) {
    free(a);
    free(bt);
    return;
  }
#pragma omp parallel for schedule(dynamic, 4)
  for (int m = 0; m < M; ++m) {
    const float* ar = a + (size_t)m * K;
    for (int n = 0; n < N; ++n) {
      const float* br = bt + (size_t)n * K;
      if (mode == 2) {
        float acc = 0.f; /* value always representable in fp16 */
        for (int k0 = 0; k0 < K; k0 += 16) {
          double part = (double)acc;
          const int ke = k0 + 16 < K ? k0 + 16 : K;
          for (int k = k0; k < ke; ++k) part += (double)ar[k] * (double)br[k];
          acc = lc_h2f(lc_d2h(part));
        }
        ((uint16_t*)C)[(size_t)m * N + n] = lc_f2h(acc);
      } else {
        double acc = 0.0; /* fp16*fp16 products are exact in fp64 */
        for (int k = 0; k < K; ++k) acc += (double)ar[k] * (double)br[k];
        if (mode == 0)
          ((uint16_t*)C)[(size_t)m * N + n] = lc_d2h(acc);
        else
          ((float*)C)[(size_t)m * N + n] = (float)acc;
      }
    }
  }
  free(a);
  free(bt);
}

void lc_oracle_hgemm_exact(const uint16_t* A, const uint16_t* B, uint16_t* C, int M, int N, int K,
                           int layout) {
  hgemm_impl(A, B, C, M, N, K, layout, 0);
}
void lc_oracle_hgemm_exact_f32(const uint16_t* A, const uint16_t* B, float* C, int M, int N, int K,
                               int layout) {
  hgemm_impl(A, B, C, M, N, K, layout, 1);
}
void lc_oracle_hgemm_refnum(const uint16_t* A, const uint16_t* B, uint16_t* C, int M, int N, int K,
                            int layout) {
  hgemm_impl(A, B, C, M, N, K, layout, 2);
}

/* fp8 OCP e4m3fn: 1 sign, 4 exponent (bias 7), 3 mantissa; 0x7f / 0xff = NaN, no infinities. */
float lc_e4m3_to_f32(uint8_t v) {
  const int sign = v >> 7, e = (v >> 3) & 0xf, m = v & 7;
  float r;
  if (e == 0xf && m == 7) return NAN;
  if (e == 0)
    r = ldexpf((float)m, -9);               /* subnormal: m/8 * 2^-6 */
  else
    r = ldexpf(1.0f + (float)m / 8.0f, e - 7);
  return sign ? -r : r;
}

void lc_oracle_gemm_fp8_exact_f32(const uint8_t* A, const uint8_t* Bnk, float* C, int M, int N, int K,
                                  float alpha) {
  float lut[256];
  for (int i = 0; i < 256; ++i) lut[i] = lc_e4m3_to_f32((uint8_t)i);
#pragma omp parallel for schedule(dynamic, 4)
  for (int m = 0; m < M; ++m) {
    const uint8_t* ar = A + (size_t)m * K;
    for (int n = 0; n < N; ++n) {
      const uint8_t* br = Bnk + (size_t)n * K;
      double acc = 0.0;
      for (int k = 0; k < K; ++k) acc += (double)lut[ar[k]] * (double)lut[br[k]];
      C[(size_t)m * N + n] = (float)(acc * (double)alpha);
    }
  }
}

/* OCP MX (microscaling) fp8: every 32 consecutive k of a row carry an E8M0 block scale, value = e4m3 * 2^(scale - 127)
 * (scale 255 = NaN).  SA [M][K/32], SB [N][K/32].  The products and the per-block partial sums are exact in fp64. */
void lc_oracle_gemm_mxfp8_exact_f32(const uint8_t* A, const uint8_t* SA, const uint8_t* Bnk, const uint8_t* SB, float* C,
                                    int M, int N, int K, float alpha) {
  float lut[256];
  double e8[256];
  const int KB = K / 32;
  for (int i = 0; i < 256; ++i) lut[i] = lc_e4m3_to_f32((uint8_t)i);
  for (int i = 0; i < 255; ++i) e8[i] = ldexp(1.0, i - 127);
  e8[255] = NAN;
#pragma omp parallel for schedule(dynamic, 4)
  for (int m = 0; m < M; ++m) {
    const uint8_t* ar = A + (size_t)m * K;
    for (int n = 0; n < N; ++n) {
      const uint8_t* br = Bnk + (size_t)n * K;
      double acc = 0.0;
      for (int kb = 0; kb < KB; ++kb) {
        double blk = 0.0;
        for (int k = 32 * kb; k < 32 * kb + 32; ++k) blk += (double)lut[ar[k]] * (double)lut[br[k]];
        acc += blk * e8[SA[(size_t)m * KB + kb]] * e8[SB[(size_t)n * KB + kb]];
      }
      C[(size_t)m * N + n] = (float)(acc * (double)alpha);
    }
  }
}

/* ---------------------------------------------------------------------------------------------- */
/* attention                                                                                       */

static inline float v_at(const float* v, int n, int d, int N, int D, int vt) {
  return vt ? v[(size_t)d * N + n] : v[(size_t)n * D + d];
}

/* Nq query rows against N keys per (b,h) problem (Nq == N for the plain entry points). */
static void attn_exact_impl(const uint16_t* Q, const uint16_t* K, const uint16_t* V, void* O, int B,
                            int H, int Nq, int N, int D, int vt, int out_f32) {
  const size_t per = (size_t)N * D;
  const size_t perq = (size_t)Nq * D;
  const double scale = 1.0 / sqrt((double)D); /* split_q.cu:79 */
  float* q = to_f32(Q, perq * B * H);
  float* k = to_f32(K, per * B * H);
  float* v = to_f32(V, per * B * H);
  if (!q || !k || !v) {
    free(q); free(k); free(v);
    return;
  }
#pragma omp parallel
  {
    double* s = (double*)malloc((size_t)N * sizeof(double));
    double* o = (double*)malloc((size_t)D * sizeof(double));
#pragma omp for collapse(2) schedule(dynamic, 8)
    for (int bh = 0; bh < B * H; ++bh) {
      for (int i = 0; i < Nq; ++i) {
        const float* qb = q + bh * perq + (size_t)i * D;
        const float* kb = k + bh * per;
        const float* vb = v + bh * per;
        double mx = -INFINITY;
        for (int j = 0; j < N; ++j) {
          double acc = 0.0;
          const float* kr = kb + (size_t)j * D;
          for (int d = 0; d < D; ++d) acc += (double)qb[d] * (double)kr[d];
          s[j] = acc * scale;
          if (s[j] > mx) mx = s[j];
        }
        double l = 0.0;
        for (int d = 0; d < D; ++d) o[d] = 0.0;
        for (int j = 0; j < N; ++j) {
          const double p = exp(s[j] - mx);
          l += p;
          if (!vt) {
            const float* vr = vb + (size_t)j * D;
            for (int d = 0; d < D; ++d) o[d] += p * (double)vr[d];
          } else {
            for (int d = 0; d < D; ++d) o[d] += p * (double)vb[(size_t)d * N + j];
          }
        }
        for (int d = 0; d < D; ++d) {
          const double r = o[d] / l;
          if (out_f32)
            ((float*)O)[bh * perq + (size_t)i * D + d] = (float)r;
          else
            ((uint16_t*)O)[bh * perq + (size_t)i * D + d] = lc_d2h(r);
        }
      }
    }
    free(s);
    free(o);
  }
  free(q); free(k); free(v);
}

void lc_oracle_attn_exact(const uint16_t* Q, const uint16_t* K, const uint16_t* V, uint16_t* O, int B,
                          int H, int N, int D, int v_transposed) {
  attn_exact_impl(Q, K, V, O, B, H, N, N, D, v_transposed, 0);
}
void lc_oracle_attn_exact_f32(const uint16_t* Q, const uint16_t* K, const uint16_t* V, float* O, int B,
                              int H, int N, int D, int v_transposed) {
  attn_exact_impl(Q, K, V, O, B, H, N, N, D, v_transposed, 1);
}
void lc_oracle_attn_exact_f32_rows(const uint16_t* Qrows, const uint16_t* K, const uint16_t* V, float* O,
                                   int BH, int Nq, int N, int D, int v_transposed) {
  attn_exact_impl(Qrows, K, V, O, BH, 1, Nq, N, D, v_transposed, 1);
}

void lc_oracle_attn_exact_f32_bf16(const uint16_t* Q, const uint16_t* K, const uint16_t* V, float* O, int B,
                                   int H, int N, int D) {
  g_decode_bf16 = 1;
  attn_exact_impl(Q, K, V, O, B, H, N, N, D, 0, 1);
  g_decode_bf16 = 0;
}

/* sampled query rows of a bf16 problem (full-size config-5 parity: (1,48,8192,512) is sampled, not swept) */
void lc_oracle_attn_exact_f32_rows_bf16(const uint16_t* Qrows, const uint16_t* K, const uint16_t* V, float* O,
                                        int BH, int Nq, int N, int D) {
  g_decode_bf16 = 1;
  attn_exact_impl(Qrows, K, V, O, BH, 1, Nq, N, D, 0, 1);
  g_decode_bf16 = 0;
}

/* fp16-accumulated dot over `len` elements in steps of 16 (one mma.sync k16 step each). */
static float dot_f16acc(const float* a, const float* b, int len, int bstride) {
  float acc = 0.f;
  for (int k0 = 0; k0 < len; k0 += 16) {
    double part = (double)acc;
    const int ke = k0 + 16 < len ? k0 + 16 : len;
    for (int k = k0; k < ke; ++k) part += (double)a[k] * (double)b[(size_t)k * bstride];
    acc = lc_h2f(lc_d2h(part));
  }
  return acc;
}

void lc_oracle_attn_refnum(const uint16_t* Q, const uint16_t* K, const uint16_t* V, uint16_t* O, int B,
                           int H, int N, int D, int Bc, int o_f32) {
  const size_t per = (size_t)N * D;
  const float scale = 1.0f / sqrtf((float)D);
  float* q = to_f32(Q, per * B * H);
  float* k = to_f32(K, per * B * H);
  float* v = to_f32(V, per * B * H);
  if (!q || !k || !v || Bc <= 0 || N % Bc != 0) {
    free(q); free(k); free(v);
    return;
  }
#pragma omp parallel
  {
    float* s = (float*)malloc((size_t)Bc * sizeof(float));
    float* p = (float*)malloc((size_t)Bc * sizeof(float));
    float* o = (float*)malloc((size_t)D * sizeof(float));
#pragma omp for collapse(2) schedule(dynamic, 8)
    for (int bh = 0; bh < B * H; ++bh) {
      for (int i = 0; i < N; ++i) {
        const float* qb = q + bh * per + (size_t)i * D;
        const float* kb = k + bh * per;
        const float* vb = v + bh * per;
        float m_old = -INFINITY, l_old = 0.f;
        for (int d = 0; d < D; ++d) o[d] = 0.f;
        for (int t = 0; t < N / Bc; ++t) {
          /* S = Q·Kᵀ, fp16 accumulate (split_q.cu:310-380), kept as fp16 values */
          float m_new = -INFINITY;
          for (int j = 0; j < Bc; ++j) {
            s[j] = dot_f16acc(qb, kb + (size_t)(t * Bc + j) * D, D, 1);
            const float sm = s[j] * scale; /* max of S*scale in fp32 (:418-435) */
            if (sm > m_new) m_new = sm;
          }
          const float m_old_eff = t > 0 ? m_old : m_new; /* :585-588 */
          if (m_old_eff > m_new) m_new = m_old_eff;      /* m = max(m_old, m_new) (:447-450) */
          /* P = exp(S*scale - m) fp32; row sum from the unrounded P (:456-468); P -> fp16 (:470-471) */
          float rs = 0.f;
          for (int j = 0; j < Bc; ++j) {
            const float e = expf(fmaf(s[j], scale, -m_new));
            rs += e;
            p[j] = rh(e);
          }
          const float resc = expf(m_old_eff - m_new); /* :590-593 */
          for (int d = 0; d < D; ++d) {
            const float pv = dot_f16acc(p, vb + (size_t)t * Bc * D + d, Bc, D); /* P·V fp16 acc (:500-560) */
            const float upd = fmaf(resc, o[d], pv);                            /* :607-610 */
            o[d] = o_f32 ? upd : rh(upd);                                      /* :611-612 vs share_qkv.cu:817 */
          }
          l_old = fmaf(resc, l_old, rs); /* :620-623 */
          m_old = m_new;
        }
        const float inv = 1.0f / l_old; /* __frcp_rn (:646-647) */
        for (int d = 0; d < D; ++d) O[bh * per + (size_t)i * D + d] = lc_f2h(inv * o[d]);
      }
    }
    free(s); free(p); free(o);
  }
  free(q); free(k); free(v);
}

/* ---------------------------------------------------------------------------------------------- */
/* bench bookkeeping restated from the reference's Python                                           */

double lc_oracle_hgemm_flops(int M, int N, int K) { return 2.0 * M * N * K; } /* hgemm.py:282 */

double lc_oracle_mha_flops(int B, int H, int N, int D, int only_matmul) { /* flash_attn_mma.py:241-278 */
  const double b = B, h = H, n = N, d = D;
  const double qk = b * h * n * n * (2 * d - 1);
  const double scaling = b * h * n * n;
  const double row_max = b * h * n * (n - 1);
  const double sub_max = b * h * n * n;
  const double ex = b * h * n * n;
  const double row_sum = b * h * n * (n - 1);
  const double norm = b * h * n * n;
  const double pv = b * h * n * d * (2 * n - 1);
  if (only_matmul) return qk + pv;
  return qk + scaling + (row_max + sub_max + ex + row_sum + norm) + pv;
}

int lc_oracle_block_swizzle_stride(int N, int K, double swizzle_factor) { /* hgemm.py:198-208 */
  if (swizzle_factor < 0) {
    swizzle_factor = N <= 4096 ? 0.5 : 0.25;
    if (N >= 14848 && K > 8192 && N % 8 == 0) swizzle_factor = 0.125;
  }
  int stride = (int)(N * swizzle_factor);
  return stride >= 256 ? stride : 1;
}

/*
 * qsae_oracle.c -- CPU restatement of the ASSERT-KTH/QuantizedSAE forward hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle for the HIP kernels in
 * quantizedsae_amd/csrc.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product path never does (it fails loudly when
 * the HIP library is missing).
 *
 * Parity status: PINNED against outputs of the reference itself, generated in the
 * build container by tools/gen_golden.py (the reference ships no tests or golden
 * vectors of its own -- SURVEY.md section 4) and committed under tests/golden/.
 *
 * Arithmetic conventions (the contract the HIP kernels are bit-exact against):
 *   - encoder contraction: latent[b][h] = fmaf-chain over k = 0..D-1 in ascending k,
 *     one rounding per step, accumulator initialised with bias[h]
 *     (reference: nn.Linear in sae/binary.py:82-84, baseline.py:8-10,
 *      ternary.py:95-98, quantized_matryoshka.py:206-209 -> F.linear/addmm).
 *     This is exactly what v_mfma_f32_32x32x2_f32 computes when K is walked in order.
 *   - top-k: the k largest by (value descending, index ascending); NaN ranks above
 *     +inf (torch.topk semantics, binary.py:94, baseline.py:35).
 *   - sparse decode: ascending-index fmaf chain over the k selected entries,
 *     then a separately rounded multiply by step and add of bias (binary.py:38).
 *   - dense decoders (ternary, matryoshka): products are exact; the sum is
 *     accumulated in double and rounded once (tolerance oracle).
 *
 * Build: oracle/Makefile  (gcc -O3 -fopenmp -ffp-contract=off -shared -fPIC)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#if defined(__x86_64__)
#define QSAE_CLONES __attribute__((target_clones("default", "avx2,fma", "avx512f")))
#else
#define QSAE_CLONES
#endif

/* fp32 cutoffs of the reference's fp32 sigmoid, s(w) = 1/(1+exp(-w)) evaluated with
 * one rounding per operation (torch CPU/CUDA both do this):
 *   s(w) >  0.5  <=>  w >= 0x33C00001 (8.9406974e-08)   (binary.py:52; latent>0.5 at
 *                                                       quantized_matryoshka.py:97)
 *   s(w) >= 0.5  <=>  w >= 0xB4400000-ish (-1.788139e-07) (quantized_matryoshka.py:70,76)
 * Measured against torch 2.10 CPU by bisection over every float (tools/gen_golden.py
 * re-checks them); tests/test_oracle_golden.py pins them. */
static inline float f32_from_bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f32_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
#define QSAE_SIGMOID_GT_HALF_BITS 0x33C00001u
#define QSAE_SIGMOID_GE_HALF_BITS 0xB43FFFFEu

float qsae_oracle_sigmoid_gt_cutoff(void) { return f32_from_bits(QSAE_SIGMOID_GT_HALF_BITS); }
float qsae_oracle_sigmoid_ge_cutoff(void) { return f32_from_bits(QSAE_SIGMOID_GE_HALF_BITS); }

static inline int sig_gt_half(float w) { return w >= f32_from_bits(QSAE_SIGMOID_GT_HALF_BITS); }
static inline int sig_ge_half(float w) { return w >= f32_from_bits(QSAE_SIGMOID_GE_HALF_BITS); }

/* ------------------------------------------------------------------------- */
/* Encoder: out[b][h] = act(bias[h] (+) sum_k x[b][k] * W[h][k]), k ascending fmaf chain.
 * act: 0 none, 1 relu, 2 sigmoid (1/(1+expf(-z)), per-op rounding; not bit-portable
 * across libm's -- callers that need exactness use act=0 and the cutoffs above). */
QSAE_CLONES
static void encode_block(const float* x, const float* Wt, const float* bias, int D, int H,
                         int h0, int hn, float* acc) {
    for (int h = 0; h < hn; ++h) acc[h] = bias ? bias[h0 + h] : 0.0f;
    for (int k = 0; k < D; ++k) {
        const float xk = x[k];
        const float* w = Wt + (size_t)k * H + h0;
        for (int h = 0; h < hn; ++h) acc[h] = __builtin_fmaf(xk, w[h], acc[h]);
    }
}

void qsae_oracle_encode(const float* x, const float* W, const float* bias, int B, int D, int H,
                        int act, float* out) {
    /* transpose W[H][D] -> Wt[D][H] so the independent per-h chains vectorise */
    float* Wt = (float*)malloc((size_t)D * H * sizeof(float));
#pragma omp parallel for schedule(static)
    for (int k = 0; k < D; ++k)
        for (int h = 0; h < H; ++h) Wt[(size_t)k * H + h] = W[(size_t)h * D + k];
    const int HB = 1024;
    const int nhb = (H + HB - 1) / HB;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b) {
        for (int hb = 0; hb < nhb; ++hb) {
            const int h0 = hb * HB, hn = (H - h0 < HB) ? (H - h0) : HB;
            float* o = out + (size_t)b * H + h0;
            encode_block(x + (size_t)b * D, Wt, bias, D, H, h0, hn, o);
            if (act == 1) { for (int h = 0; h < hn; ++h) o[h] = o[h] > 0.0f ? o[h] : 0.0f; }
            else if (act == 2) { for (int h = 0; h < hn; ++h) o[h] = 1.0f / (1.0f + expf(-o[h])); }
        }
    }
    free(Wt);
}

/* Exact single dot: the same chain, scalar (used to cross-check the blocked loop). */
float qsae_oracle_dot_chain(const float* x, const float* w, float bias, int D) {
    float acc = bias;
    for (int k = 0; k < D; ++k) acc = fmaf(x[k], w[k], acc);
    return acc;
}

/* ------------------------------------------------------------------------- */
/* Total order for top-k: larger key wins.  key = (monotone(value) << 32) | ~index. */
static inline uint32_t mono_key(float v) {
    if (v != v) return 0xFFFFFFFFu;               /* NaN above everything (torch.topk) */
    uint32_t u = f32_bits(v);
    if (u == 0x80000000u) u = 0;                  /* -0 == +0 */
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
static inline uint64_t full_key(float v, uint32_t idx) {
    return ((uint64_t)mono_key(v) << 32) | (uint32_t)(~idx);
}
static int cmp_key_desc(const void* a, const void* b) {
    uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
    return (x < y) - (x > y);
}
static int cmp_i32_asc(const void* a, const void* b) {
    int32_t x = *(const int32_t*)a, y = *(const int32_t*)b;
    return (x > y) - (x < y);
}

/* idx/val [B][k], ordered by (value desc, index asc).  k <= H. */
void qsae_oracle_topk(const float* latent, int B, int H, int k, int32_t* idx, float* val) {
#pragma omp parallel
    {
        uint64_t* keys = (uint64_t*)malloc((size_t)H * sizeof(uint64_t));
#pragma omp for schedule(static)
        for (int b = 0; b < B; ++b) {
            const float* row = latent + (size_t)b * H;
            for (int h = 0; h < H; ++h) keys[h] = full_key(row[h], (uint32_t)h);
            qsort(keys, (size_t)H, sizeof(uint64_t), cmp_key_desc);
            for (int j = 0; j < k; ++j) {
                int32_t h = (int32_t)(~(uint32_t)(keys[j] & 0xFFFFFFFFu));
                idx[(size_t)b * k + j] = h;
                val[(size_t)b * k + j] = row[h];
            }
        }
        free(keys);
    }
}

/* gap[b] = value(k-th) - value((k+1)-th), for the near-tie audit. */
void qsae_oracle_topk_gap(const float* latent, int B, int H, int k, float* gap) {
#pragma omp parallel
    {
        uint64_t* keys = (uint64_t*)malloc((size_t)H * sizeof(uint64_t));
#pragma omp for schedule(static)
        for (int b = 0; b < B; ++b) {
            const float* row = latent + (size_t)b * H;
            for (int h = 0; h < H; ++h) keys[h] = full_key(row[h], (uint32_t)h);
            qsort(keys, (size_t)H, sizeof(uint64_t), cmp_key_desc);
            if (k < H) {
                int32_t a = (int32_t)(~(uint32_t)(keys[k - 1] & 0xFFFFFFFFu));
                int32_t c = (int32_t)(~(uint32_t)(keys[k] & 0xFFFFFFFFu));
                gap[b] = row[a] - row[c];
            } else gap[b] = INFINITY;
        }
        free(keys);
    }
}

/* dense[b][h] = val if selected else +0  (binary.py:96-99, baseline.py:38-40) */
void qsae_oracle_densify(const int32_t* idx, const float* val, int B, int k, int H, float* dense) {
    memset(dense, 0, (size_t)B * H * sizeof(float));
    for (int b = 0; b < B; ++b)
        for (int j = 0; j < k; ++j) dense[(size_t)b * H + idx[(size_t)b * k + j]] = val[(size_t)b * k + j];
}

/* ------------------------------------------------------------------------- */
/* BinarySAE decoder packer (binary.py:49-58): hard bit = sigmoid(logit) > 0.5,
 * column d*n+b is bit b (LSB first) of output d, MSB weight negative (two's complement).
 * Storage: field width fw = 1,2,4,8 (smallest power of two >= n), fields little-endian
 * inside each byte, row h contiguous: packed[h][row_bytes], row_bytes = D*fw/8 rounded up
 * to a multiple of 4.  The field holds the n-bit two's-complement code (upper fw-n bits zero). */
static int field_width(int n) { return n <= 1 ? 1 : n <= 2 ? 2 : n <= 4 ? 4 : 8; }
int qsae_oracle_binary_row_bytes(int D, int n) { return ((D * field_width(n) + 31) / 32) * 4; }

void qsae_oracle_pack_binary(const float* logits, int H, int D, int n, uint8_t* packed) {
    const int fw = field_width(n), rb = qsae_oracle_binary_row_bytes(D, n);
    memset(packed, 0, (size_t)H * rb);
#pragma omp parallel for schedule(static)
    for (int h = 0; h < H; ++h) {
        const float* lr = logits + (size_t)h * D * n;
        uint8_t* pr = packed + (size_t)h * rb;
        for (int d = 0; d < D; ++d) {
            unsigned code = 0;
            for (int b = 0; b < n; ++b) code |= (unsigned)sig_gt_half(lr[d * n + b]) << b;
            const int bitpos = d * fw;
            pr[bitpos >> 3] |= (uint8_t)(code << (bitpos & 7));
        }
    }
}

static inline int binary_weight(const uint8_t* pr, int d, int n, int fw) {
    const int bitpos = d * fw;
    unsigned code = (pr[bitpos >> 3] >> (bitpos & 7)) & ((1u << fw) - 1u);
    code &= (1u << n) - 1u;
    int w = (int)code;
    if (code & (1u << (n - 1))) w -= (1 << n);   /* sign-extend from n bits */
    return w;
}

/* int_weights[h][d] as float (binary.py:49-58) -- for decoder_dictionary parity. */
void qsae_oracle_unpack_binary(const uint8_t* packed, int H, int D, int n, float* w) {
    const int fw = field_width(n), rb = qsae_oracle_binary_row_bytes(D, n);
    for (int h = 0; h < H; ++h)
        for (int d = 0; d < D; ++d) w[(size_t)h * D + d] = (float)binary_weight(packed + (size_t)h * rb, d, n, fw);
}

/* polarize_loss = mean(p (1-p) 2^b) over all logits (binary.py:42-43), double accumulate. */
double qsae_oracle_polarize(const float* logits, int H, int D, int n) {
    double s = 0.0;
    const size_t total = (size_t)H * D * n;
#pragma omp parallel for reduction(+ : s) schedule(static)
    for (size_t i = 0; i < total; ++i) {
        const float p = 1.0f / (1.0f + expf(-logits[i]));
        s += (double)(p * (1.0f - p) * (float)(1 << (i % (size_t)n)));
    }
    return s / (double)total;
}

/* recon[b][d] = step * (sum_j val_j * w[idx_j][d]) + bias[d]; j in ascending idx, fmaf chain;
 * multiply and add rounded separately (binary.py:38). */
void qsae_oracle_decode_binary(const int32_t* idx, const float* val, int B, int k,
                               const uint8_t* packed, int D, int n, float step,
                               const float* bias, float* recon) {
    const int fw = field_width(n), rb = qsae_oracle_binary_row_bytes(D, n);
#pragma omp parallel
    {
        int32_t* order = (int32_t*)malloc((size_t)k * 2 * sizeof(int32_t));
#pragma omp for schedule(static)
        for (int b = 0; b < B; ++b) {
            /* sort (idx, position) ascending by idx */
            int64_t* pr = (int64_t*)malloc((size_t)k * sizeof(int64_t));
            for (int j = 0; j < k; ++j) pr[j] = ((int64_t)idx[(size_t)b * k + j] << 32) | (uint32_t)j;
            for (int i = 1; i < k; ++i) { int64_t t = pr[i]; int j = i - 1; while (j >= 0 && pr[j] > t) { pr[j + 1] = pr[j]; --j; } pr[j + 1] = t; }
            for (int d = 0; d < D; ++d) {
                float acc = 0.0f;
                for (int j = 0; j < k; ++j) {
                    const int h = (int)(pr[j] >> 32), p = (int)(pr[j] & 0xFFFFFFFF);
                    const float w = (float)binary_weight(packed + (size_t)h * rb, d, n, fw);
                    acc = fmaf(val[(size_t)b * k + p], w, acc);
                }
                float r = step * acc;
                r = r + (bias ? bias[d] : 0.0f);
                recon[(size_t)b * D + d] = r;
            }
            free(pr);
        }
        free(order);
    }
}

/* Same chain with an fp32 table[H][D] (baseline decoder transposed, baseline.py:29;
 * BinarySAE "soft" int_weights, binary.py:26-38).  recon = scale*acc + bias with the
 * multiply skipped when scale == 1 (baseline has none). */
void qsae_oracle_decode_table(const int32_t* idx, const float* val, int B, int k,
                              const float* table, int D, float scale, const float* bias,
                              float* recon) {
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) {
        int64_t* pr = (int64_t*)malloc((size_t)k * sizeof(int64_t));
        for (int j = 0; j < k; ++j) pr[j] = ((int64_t)idx[(size_t)b * k + j] << 32) | (uint32_t)j;
        for (int i = 1; i < k; ++i) { int64_t t = pr[i]; int j = i - 1; while (j >= 0 && pr[j] > t) { pr[j + 1] = pr[j]; --j; } pr[j + 1] = t; }
        for (int d = 0; d < D; ++d) {
            float acc = 0.0f;
            for (int j = 0; j < k; ++j) {
                const int h = (int)(pr[j] >> 32), p = (int)(pr[j] & 0xFFFFFFFF);
                acc = fmaf(val[(size_t)b * k + p], table[(size_t)h * D + d], acc);
            }
            float r = (scale == 1.0f) ? acc : scale * acc;
            r = r + (bias ? bias[d] : 0.0f);
            recon[(size_t)b * D + d] = r;
        }
        free(pr);
    }
}

/* ------------------------------------------------------------------------- */
/* Ternary decoder (ternary.py:41-52): hard[d][h] = sign(w) * (|w| >= 0.5); no bias.
 * codes[d][h] int8 in {-1,0,1}. */
void qsae_oracle_ternary_codes(const float* w, size_t n, int8_t* codes) {
    for (size_t i = 0; i < n; ++i) {
        const float a = fabsf(w[i]);
        codes[i] = (a >= 0.5f) ? (w[i] > 0.0f ? 1 : -1) : 0;   /* NaN -> 0 (|NaN|>=0.5 false) */
    }
}
void qsae_oracle_decode_ternary(const float* h, int B, int H, const int8_t* codes, int D, float* recon) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int d = 0; d < D; ++d) {
            double acc = 0.0;
            const float* hr = h + (size_t)b * H;
            const int8_t* cr = codes + (size_t)d * H;
            for (int j = 0; j < H; ++j) acc += (double)hr[j] * (double)cr[j];
            recon[(size_t)b * D + d] = (float)acc;
        }
}

/* ------------------------------------------------------------------------- */
/* Matryoshka (quantized_matryoshka.py:25-38): nested level sizes. */
void qsae_oracle_matryoshka_sizes(int H, int n, int32_t* sizes) {
    long sum = 0;
    for (int i = 0; i < n; ++i) { sizes[i] = (i < 2) ? 1 : (1 << (i - 1)); sum += sizes[i]; }
    if (sum != H) {
        const double sf = (double)H / (double)sum;
        long acc = 0;
        for (int i = 0; i < n; ++i) { int s = (int)((double)sizes[i] * sf); if (s < 1) s = 1; sizes[i] = s; }
        for (int i = 0; i < n - 1; ++i) acc += sizes[i];
        sizes[n - 1] = (int32_t)(H - acc);
    }
}

/* codes[j][d] = sgn(sig(w)>=.5) + sgn(sig(wm)>=.5) in {-2,0,2}  (:67-80)
 * scale[j]   = reciprocal(||S_j||_2 + 1e-8) * (2^(n-i-2) * quant_step)  (:82-90; torch's
 *              `float / tensor` is reciprocal()*float) for j in level i. */
void qsae_oracle_matryoshka_pack(const float* w, const float* wm, int H, int D, int n,
                                 float abs_range, int8_t* codes, float* scale) {
    int32_t sizes[32];
    qsae_oracle_matryoshka_sizes(H, n, sizes);
    const double quant_step = (double)abs_range / ldexp(1.0, n - 1);
    int start = 0;
    for (int i = 0; i < n; ++i) {
        const float sf = (float)(ldexp(1.0, n - i - 2) * quant_step);
        for (int j = start; j < start + sizes[i]; ++j) {
            int nz = 0;
            for (int d = 0; d < D; ++d) {
                const int s = (sig_ge_half(w[(size_t)j * D + d]) ? 1 : -1) + (sig_ge_half(wm[(size_t)j * D + d]) ? 1 : -1);
                codes[(size_t)j * D + d] = (int8_t)s;
                nz += (s != 0);
            }
            const float norm = sqrtf((float)(4 * nz));
            const float denom = norm + 1e-8f;
            const float rcp = 1.0f / denom;
            scale[j] = rcp * sf;
        }
        start += sizes[i];
    }
}

/* zbits[b][j] in {0,1}; levels[i][b][d] cumulative reconstructions (:121-129);
 * l0[i] = mean_b sum_j z (:128).  Per-level partial sums in double, rounded to
 * fp32 once, then the fp32 running sum and the level-0 bias add as the reference. */
void qsae_oracle_decode_matryoshka(const uint8_t* zbits, int B, int H, int D, int n,
                                   const int8_t* codes, const float* scale, const float* bias,
                                   int allow_bias, float* levels, float* l0) {
    int32_t sizes[32];
    qsae_oracle_matryoshka_sizes(H, n, sizes);
    float* recon = (float*)calloc((size_t)B * D, sizeof(float));
    int start = 0;
    for (int i = 0; i < n; ++i) {
        const int sz = sizes[i];
        double cnt = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : cnt)
        for (int b = 0; b < B; ++b) {
            double* acc = (double*)calloc((size_t)D, sizeof(double));
            const uint8_t* zr = zbits + (size_t)b * H + start;
            for (int j = 0; j < sz; ++j) {
                if (!zr[j]) continue;
                cnt += 1.0;
                const double s = (double)scale[start + j];
                const int8_t* cr = codes + (size_t)(start + j) * D;
                for (int d = 0; d < D; ++d) acc[d] += s * (double)cr[d];
            }
            for (int d = 0; d < D; ++d) {
                float r = recon[(size_t)b * D + d] + (float)acc[d];
                if (i == 0 && allow_bias && bias) r = r + bias[d];
                recon[(size_t)b * D + d] = r;
            }
            free(acc);
        }
        memcpy(levels + (size_t)i * B * D, recon, (size_t)B * D * sizeof(float));
        l0[i] = (float)(cnt / (double)B);
        start += sz;
    }
    free(recon);
}

/* z bit from a pre-activation: sigmoid(z) > 0.5 (quantized_matryoshka.py:97,209). */
void qsae_oracle_zbits(const float* pre, size_t n, uint8_t* bits) {
    for (size_t i = 0; i < n; ++i) bits[i] = (uint8_t)sig_gt_half(pre[i]);
}

/* ------------------------------------------------------------------------- */
/* Sum of squared error (dynamic_analysis.py:86-100): fp32 diff, fp32 square, double sum. */
double qsae_oracle_sq_err_sum(const float* recon, const float* x, size_t n) {
    double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
    for (size_t i = 0; i < n; ++i) { const float d = recon[i] - x[i]; s += (double)(d * d); }
    return s;
}

int qsae_oracle_num_threads(void) {
    int n = 1;
#ifdef _OPENMP
    extern int omp_get_max_threads(void);
    n = omp_get_max_threads();
#endif
    return n;
}

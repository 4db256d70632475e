/*
 * es_oracle.c -- CPU restatement of the EfficientSpeech acoustic-model forward path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle for the HIP path: it may be
 * built, loaded and called only by tests/, __graft_entry__.smoke() and the `cpu_baseline`
 * leg of bench.py -- never by the product path (efficientspeech_amd/), which fails loudly
 * when libesmi.so (the HIP extension) is missing.
 *
 * Parity pinning: the reference (/root/reference, pure PyTorch, no tests / golden vectors
 * of its own -- SURVEY.md §4) is importable in the build container.  tools/gen_golden.py runs
 * the reference modules on seeded weights/inputs and commits the results under tests/golden/;
 * tests/test_oracle_golden.py checks every function below against those vectors.  So the
 * oracle is pinned to outputs of the reference itself run in the build container.
 *
 * Every function cites the reference lines it restates (paths relative to /root/reference).
 * All tensors are channels-last (B, N, C) contiguous float32; dot products accumulate in
 * `acc_t` (double by default = a tighter anchor than either fp32 implementation; build with
 * -DESO_FLOAT_ACC for the fp32-accumulating variant used as the timed CPU baseline).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef ESO_FLOAT_ACC
typedef float acc_t;
#else
typedef double acc_t;
#endif

#define ESO_OK 0
#define ESO_ERR_MISSING_WEIGHT -2
#define ESO_ERR_SHAPE -3
#define ESO_ERR_ARG -4

typedef struct {
    int depth, reduction, head, embed_dim, kernel_size, expansion;
    int n_blocks, block_depth, dec_kernel, n_mel, vocab; /* vocab = len(symbols)+1 = 153 */
} eso_cfg;

/* name -> pointer table handed over by the Python wrapper (checkpoint key names) */
typedef struct {
    int n;
    const char* const* names;
    const float* const* ptrs;
} eso_weights;

static const float* W(const eso_weights* w, const char* fmt, int a, int b, int* err) {
    char key[160];
    snprintf(key, sizeof key, fmt, a, b);
    for (int i = 0; i < w->n; ++i)
        if (strcmp(w->names[i], key) == 0) return w->ptrs[i];
    fprintf(stderr, "es_oracle: missing weight %s\n", key);
    *err = ESO_ERR_MISSING_WEIGHT;
    return NULL;
}

int eso_acc_bytes(void) { return (int)sizeof(acc_t); }

/* ---------------------------------------------------------------- primitives */

/* nn.Linear: y[r][n] = sum_k x[r][k] W[n][k] + b[n]   (torch F.linear; W is (N,K)).
 * W is transposed once per call so that the inner loop runs over contiguous output channels (vectorises
 * without horizontal reductions); the accumulation order over k is the natural one. */
static void linear(const float* x, long rows, int K, const float* Wt, const float* b, int N, float* y) {
    acc_t* wT = (acc_t*)malloc(sizeof(acc_t) * (size_t)K * N);
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) wT[(size_t)k * N + n] = (acc_t)Wt[(size_t)n * K + k];
#pragma omp parallel
    {
        acc_t* t = (acc_t*)malloc(sizeof(acc_t) * (size_t)N);
#pragma omp for schedule(static)
        for (long r = 0; r < rows; ++r) {
            const float* xr = x + r * K;
            for (int n = 0; n < N; ++n) t[n] = b ? (acc_t)b[n] : (acc_t)0;
            for (int k = 0; k < K; ++k) {
                const acc_t xk = (acc_t)xr[k];
                const acc_t* wk = wT + (size_t)k * N;
                for (int n = 0; n < N; ++n) t[n] += xk * wk[n];
            }
            for (int n = 0; n < N; ++n) y[r * N + n] = (float)t[n];
        }
        free(t);
    }
    free(wT);
}

/* nn.Conv1d on channels-last data: cross-correlation, zero padding, weight (Cout,Cin,k).
 * y[b][t][co] = bias[co] + sum_{j,ci} W[co][ci][j] * x[b][t*stride + j - pad][ci]          */
static int conv_out_len(int n, int k, int stride, int pad) { return (n + 2 * pad - k) / stride + 1; }

static void conv1d_cl(const float* x, int B, int Nin, int Cin, const float* Wt, const float* bias, int Cout,
                      int k, int stride, int pad, float* y) {
    const int Nout = conv_out_len(Nin, k, stride, pad);
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < Nout; ++t) {
            float* yr = y + ((long)b * Nout + t) * Cout;
            for (int co = 0; co < Cout; ++co) {
                acc_t s = bias ? (acc_t)bias[co] : (acc_t)0;
                for (int j = 0; j < k; ++j) {
                    const int ti = t * stride + j - pad;
                    if (ti < 0 || ti >= Nin) continue;
                    const float* xr = x + ((long)b * Nin + ti) * Cin;
                    const float* wr = Wt + (long)co * Cin * k + j;
                    for (int ci = 0; ci < Cin; ++ci) s += (acc_t)xr[ci] * (acc_t)wr[(long)ci * k];
                }
                yr[co] = (float)s;
            }
        }
}

/* depthwise nn.Conv1d(groups=C): weight (C,1,k) */
static void dwconv1d_cl(const float* x, int B, int N, int C, const float* Wt, const float* bias, int k, float* y) {
    const int pad = k / 2;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < N; ++t) {
            float* yr = y + ((long)b * N + t) * C;
            for (int c = 0; c < C; ++c) {
                acc_t s = (acc_t)bias[c];
                for (int j = 0; j < k; ++j) {
                    const int ti = t + j - pad;
                    if (ti < 0 || ti >= N) continue;
                    s += (acc_t)x[((long)b * N + ti) * C + c] * (acc_t)Wt[c * k + j];
                }
                yr[c] = (float)s;
            }
        }
}

/* nn.LayerNorm(C): eps 1e-5, biased variance, affine; in place over `rows` rows */
static void layernorm(float* x, long rows, int C, const float* g, const float* b) {
#pragma omp parallel for schedule(static)
    for (long r = 0; r < rows; ++r) {
        float* xr = x + r * C;
        acc_t m = 0, v = 0;
        for (int c = 0; c < C; ++c) m += (acc_t)xr[c];
        m /= (acc_t)C;
        for (int c = 0; c < C; ++c) {
            acc_t d = (acc_t)xr[c] - m;
            v += d * d;
        }
        v /= (acc_t)C;
        const acc_t rs = (acc_t)1 / (acc_t)sqrt((double)v + 1e-5);
        for (int c = 0; c < C; ++c) xr[c] = (float)(((acc_t)xr[c] - m) * rs * (acc_t)g[c] + (acc_t)b[c]);
    }
}

static void add_inplace(float* y, const float* x, long n) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i) y[i] = y[i] + x[i];
}

/* x.masked_fill(mask[...,None], 0): zero rows where mask (B*N bytes, 1 = padding) is set */
static void mask_rows(float* x, const uint8_t* mask, long rows, int C) {
    if (!mask) return;
    for (long r = 0; r < rows; ++r)
        if (mask[r]) memset(x + r * C, 0, sizeof(float) * C);
}

static float gelu_erf(float v) { return (float)(0.5 * (double)v * (1.0 + erf((double)v * 0.70710678118654752440))); }

/* torch.round: half to even.  rint() under the default FE_TONEAREST mode. */
static float round_half_even(float v) { return (float)rint((double)v); }

/* ---------------------------------------------------------------- SelfAttention
 * layers/blocks.py:43-71.  qkv = Linear(C, 3*h*C, bias=False) reshaped (B,N,3,h,C): the
 * output channel index is s*h*C + hd*C + c.  Every head uses the FULL width C;
 * scale = (C // h) ** -0.5 (:37-38).  NO mask is applied to the scores (:59-63 build
 * attn_mask but never use it) -- padded keys take part.  out = Linear(h*C, C)+bias of the
 * heads concatenated head-major.                                                         */
static void self_attention(const float* x, int B, int N, int C, int h, const float* Wqkv, const float* Wproj,
                           const float* bproj, float* y) {
    const int HC = h * C;
    float* qkv = (float*)malloc(sizeof(float) * (size_t)B * N * 3 * HC);
    float* ctx = (float*)malloc(sizeof(float) * (size_t)B * N * HC);
    linear(x, (long)B * N, C, Wqkv, NULL, 3 * HC, qkv);
    const double scale = pow((double)(C / h), -0.5);
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int hd = 0; hd < h; ++hd) {
            acc_t* p = (acc_t*)malloc(sizeof(acc_t) * N);
            for (int i = 0; i < N; ++i) {
                const float* q = qkv + ((long)(b * N + i) * 3 + 0) * HC + hd * C;
                acc_t mx = -INFINITY;
                for (int j = 0; j < N; ++j) {
                    const float* kk = qkv + ((long)(b * N + j) * 3 + 1) * HC + hd * C;
                    acc_t s = 0;
                    for (int c = 0; c < C; ++c) s += (acc_t)q[c] * (acc_t)kk[c];
                    s = (acc_t)((float)s * (float)scale); /* fp32 product then scale, blocks.py:50 */
                    p[j] = s;
                    if (s > mx) mx = s;
                }
                acc_t den = 0;
                for (int j = 0; j < N; ++j) {
                    p[j] = (acc_t)exp((double)(p[j] - mx));
                    den += p[j];
                }
                float* o = ctx + (long)(b * N + i) * HC + hd * C;
                for (int c = 0; c < C; ++c) {
                    acc_t s = 0;
                    for (int j = 0; j < N; ++j)
                        s += (acc_t)(float)(p[j] / den) * (acc_t)qkv[((long)(b * N + j) * 3 + 2) * HC + hd * C + c];
                    o[c] = (float)s;
                }
            }
            free(p);
        }
    linear(ctx, (long)B * N, HC, Wproj, bproj, C, y);
    free(qkv);
    free(ctx);
}

/* ---------------------------------------------------------------- MixFFN
 * layers/blocks.py:22-29: Linear(C,eC) -> dense Conv1d(eC,eC,3,pad 1) -> GELU(erf) -> Linear(eC,C) */
static void mixffn(const float* x, int B, int N, int C, int e, const float* w1, const float* b1, const float* wc,
                   const float* bc, const float* w2, const float* b2, float* y) {
    const int E = C * e;
    float* t1 = (float*)malloc(sizeof(float) * (size_t)B * N * E);
    float* t2 = (float*)malloc(sizeof(float) * (size_t)B * N * E);
    linear(x, (long)B * N, C, w1, b1, E, t1);
    conv1d_cl(t1, B, N, E, wc, bc, E, 3, 1, 1, t2);
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)B * N * E; ++i) t2[i] = gelu_erf(t2[i]);
    linear(t2, (long)B * N, E, w2, b2, C, y);
    free(t1);
    free(t2);
}

/* the two sub-module forwards on their own (tests of the module-level API; weights in checkpoint layouts) */
int eso_self_attention(int B, int N, int C, int h, const float* x, const float* Wqkv, const float* Wproj,
                       const float* bproj, float* y) {
    self_attention(x, B, N, C, h, Wqkv, Wproj, bproj, y);
    return ESO_OK;
}
int eso_mixffn(int B, int N, int C, int e, const float* x, const float* w1, const float* b1, const float* wc,
               const float* bc, const float* w2, const float* b2, float* y) {
    mixffn(x, B, N, C, e, w1, b1, wc, bc, w2, b2, y);
    return ESO_OK;
}

/* pooled padding mask of an encoder block: blocks.py:51-57.
 * pad the (B,T) mask with True to a multiple of `pool`, then max over groups of `pool`. */
static void pool_mask(const uint8_t* mask, int B, int T, int pool, uint8_t* out, int Nout) {
    for (int b = 0; b < B; ++b)
        for (int n = 0; n < Nout; ++n) {
            uint8_t m = 0;
            for (int p = 0; p < pool; ++p) {
                const int t = n * pool + p;
                m |= (t >= T) ? 1 : mask[b * T + t];
            }
            out[b * Nout + n] = m;
        }
}

/* sequence length after encoder block i (stride 1 for block 0, 2 afterwards; networks.py:27-30) */
int eso_block_len(const eso_cfg* c, int T, int blk) {
    int n = T;
    for (int i = 0; i <= blk; ++i) {
        const int k = c->kernel_size - (i > 0 ? 2 : 0);
        n = conv_out_len(n, k, i > 0 ? 2 : 1, k / 2);
    }
    return n;
}

/* ---------------------------------------------------------------- Encoder.forward
 * layers/networks.py:52-87.  phoneme int32 (B,T); mask uint8 (B,T) or NULL (B==1 path,
 * networks.py:338).  feats[i] receives block i's output (B,N_i,dim*2^i); bmask[i] (B,N_i)
 * the pooled mask of block i (bmask[0] is `decoder_mask`, :76-77).                        */
int eso_encoder(const eso_cfg* c, const eso_weights* w, int B, int T, const int32_t* phoneme, const uint8_t* mask,
                float* const* feats, uint8_t* const* bmask) {
    int err = ESO_OK;
    const int E = c->embed_dim, dim = E / c->reduction;
    const float* emb = W(w, "encoder.encoder.embed.weight", 0, 0, &err);
    if (err) return err;
    float* x = (float*)malloc(sizeof(float) * (size_t)B * T * E);
    for (long r = 0; r < (long)B * T; ++r) { /* nn.Embedding gather, :54 */
        const int id = phoneme[r];
        if (id < 0 || id >= c->vocab) { free(x); return ESO_ERR_ARG; }
        memcpy(x + r * E, emb + (long)id * E, sizeof(float) * E);
    }
    int n_in = T, c_in = E;
    for (int i = 0; i < c->depth && !err; ++i) {
        const int c_out = dim << i, h = c->head * (i + 1);
        const int k = c->kernel_size - (i > 0 ? 2 : 0), stride = i > 0 ? 2 : 1;
        const int n = conv_out_len(n_in, k, stride, k / 2);
        const float* w0 = W(w, "encoder.encoder.attn_blocks.%d.0.weight", i, 0, &err);
        const float* w1 = W(w, "encoder.encoder.attn_blocks.%d.1.weight", i, 0, &err);
        const float* wqkv = W(w, "encoder.encoder.attn_blocks.%d.2.qkv.weight", i, 0, &err);
        const float* wproj = W(w, "encoder.encoder.attn_blocks.%d.2.proj.weight", i, 0, &err);
        const float* bproj = W(w, "encoder.encoder.attn_blocks.%d.2.proj.bias", i, 0, &err);
        const float* m1w = W(w, "encoder.encoder.attn_blocks.%d.3.mlp1.weight", i, 0, &err);
        const float* m1b = W(w, "encoder.encoder.attn_blocks.%d.3.mlp1.bias", i, 0, &err);
        const float* mcw = W(w, "encoder.encoder.attn_blocks.%d.3.conv.weight", i, 0, &err);
        const float* mcb = W(w, "encoder.encoder.attn_blocks.%d.3.conv.bias", i, 0, &err);
        const float* m2w = W(w, "encoder.encoder.attn_blocks.%d.3.mlp2.weight", i, 0, &err);
        const float* m2b = W(w, "encoder.encoder.attn_blocks.%d.3.mlp2.bias", i, 0, &err);
        const float* n1g = W(w, "encoder.encoder.attn_blocks.%d.4.weight", i, 0, &err);
        const float* n1b = W(w, "encoder.encoder.attn_blocks.%d.4.bias", i, 0, &err);
        const float* n2g = W(w, "encoder.encoder.attn_blocks.%d.5.weight", i, 0, &err);
        const float* n2b = W(w, "encoder.encoder.attn_blocks.%d.5.bias", i, 0, &err);
        if (err) break;
        /* merge convs :64-67 -- the "3x3" conv is DENSE (no groups=), bias=False (:40-42) */
        float* t = (float*)malloc(sizeof(float) * (size_t)B * n * c_in);
        conv1d_cl(x, B, n_in, c_in, w0, NULL, c_in, k, stride, k / 2, t);
        float* xo = feats[i];
        conv1d_cl(t, B, n, c_in, w1, NULL, c_out, 1, 1, 0, xo);
        free(t);
        /* pooled mask :69-70 + blocks.py:51-57 */
        const uint8_t* bm = NULL;
        if (mask) {
            const int pool = (int)round_half_even((float)T / (float)n);
            const int npool = (T + pool - 1) / pool;
            if (npool != n) err = ESO_ERR_SHAPE; /* the reference would fail to broadcast here */
            else { pool_mask(mask, B, T, pool, bmask[i], n); bm = bmask[i]; }
            if (err) break;
        }
        const long rows = (long)B * n;
        float* y = (float*)malloc(sizeof(float) * (size_t)rows * c_out);
        self_attention(xo, B, n, c_out, h, wqkv, wproj, bproj, y); /* :72 */
        add_inplace(y, xo, rows * c_out);                          /* :73 norm1(y + x) */
        layernorm(y, rows, c_out, n1g, n1b);
        mask_rows(y, bm, rows, c_out); /* :74-75 */
        mixffn(y, B, n, c_out, c->expansion, m1w, m1b, mcw, mcb, m2w, m2b, xo);
        add_inplace(xo, y, rows * c_out); /* :80 norm2(mixffn(x) + x) */
        layernorm(xo, rows, c_out, n2g, n2b);
        mask_rows(xo, bm, rows, c_out); /* :82-83 */
        free(y);
        free(x);
        x = (float*)malloc(sizeof(float) * (size_t)rows * c_out);
        memcpy(x, xo, sizeof(float) * (size_t)rows * c_out);
        n_in = n;
        c_in = c_out;
    }
    free(x);
    return err;
}

/* ---------------------------------------------------------------- Fuse.forward
 * layers/networks.py:189-219.  Linear(dim*2^i, dim); for i>=1 a DENSE ConvTranspose1d
 * (weight (Cin,Cout,k), stride 2^i, no padding: y[p] += x[n] W[:, :, j] for p = n*s + j),
 * cropped to T (:203-206); channel concat; Linear(depth*dim, dim); masked_fill.          */
int eso_fuse(const eso_cfg* c, const eso_weights* w, int B, int T, const float* const* feats, const uint8_t* mask0,
             float* fused) {
    int err = ESO_OK;
    const int dim = c->embed_dim / c->reduction, D = c->depth;
    const long rows = (long)B * T;
    float* cat = (float*)calloc((size_t)rows * dim * D, sizeof(float));
    for (int i = 0; i < D && !err; ++i) {
        const int n = eso_block_len(c, T, i), ci = dim << i, s = 1 << i, k = c->kernel_size;
        const float* lw = W(w, "encoder.fuse.mlps.%d.0.weight", i, 0, &err);
        const float* lb = W(w, "encoder.fuse.mlps.%d.0.bias", i, 0, &err);
        if (err) break;
        float* t = (float*)malloc(sizeof(float) * (size_t)B * n * dim);
        linear(feats[i], (long)B * n, ci, lw, lb, dim, t);
        if (i == 0) {
            for (long r = 0; r < rows; ++r) memcpy(cat + r * dim * D, t + r * dim, sizeof(float) * dim);
        } else {
            const float* tw = W(w, "encoder.fuse.mlps.%d.1.weight", i, 0, &err);
            const float* tb = W(w, "encoder.fuse.mlps.%d.1.bias", i, 0, &err);
            if (err) { free(t); break; }
            if ((n - 1) * s + k < T) { free(t); err = ESO_ERR_SHAPE; break; } /* torch.cat would fail */
            for (int b = 0; b < B; ++b)
                for (int p = 0; p < T; ++p)
                    for (int co = 0; co < dim; ++co) {
                        acc_t a = (acc_t)tb[co];
                        for (int j = 0; j < k; ++j) {
                            if ((p - j) % s != 0 || p - j < 0) continue;
                            const int nn = (p - j) / s;
                            if (nn >= n) continue;
                            const float* xr = t + ((long)b * n + nn) * dim;
                            for (int cc = 0; cc < dim; ++cc) a += (acc_t)xr[cc] * (acc_t)tw[((long)cc * dim + co) * k + j];
                        }
                        cat[((long)b * T + p) * dim * D + i * dim + co] = (float)a;
                    }
        }
        free(t);
    }
    if (!err) {
        const float* fw = W(w, "encoder.fuse.fuse.weight", 0, 0, &err);
        const float* fb = W(w, "encoder.fuse.fuse.bias", 0, 0, &err);
        if (!err) {
            linear(cat, rows, dim * D, fw, fb, dim, fused);
            mask_rows(fused, mask0, rows, dim); /* :216-217 */
        }
    }
    free(cat);
    return err;
}

/* ---------------------------------------------------------------- AcousticDecoder.forward
 * layers/networks.py:151-165.  which: 0 pitch, 1 energy, 2 duration.
 * y = ReLU(conv1(x)); y = ReLU(LN1(y)); y = ReLU(conv2(y)); features = LN2(y);
 * pred = Linear(dim,1)(y)  -- on the PRE-norm2 tensor (:157-160); duration adds ReLU (:161-163).
 * `features` may be NULL (pitch/energy discard it).                                        */
int eso_acoustic(const eso_cfg* c, const eso_weights* w, int which, int B, int T, const float* fused, float* pred,
                 float* features) {
    static const char* nm[3] = {"pitch", "energy", "duration"};
    int err = ESO_OK;
    const int dim = c->embed_dim / c->reduction;
    char f[8][128];
    const char* sfx[8] = {"conv1.0.weight", "conv1.0.bias", "norm1.weight", "norm1.bias",
                          "conv2.0.weight", "conv2.0.bias", "norm2.weight", "norm2.bias"};
    const float* p[8];
    for (int i = 0; i < 8; ++i) {
        snprintf(f[i], sizeof f[i], "encoder.%s_decoder.%s", nm[which], sfx[i]);
        p[i] = W(w, f[i], 0, 0, &err);
    }
    char lf[2][128];
    snprintf(lf[0], sizeof lf[0], "encoder.%s_decoder.linear.weight", nm[which]);
    snprintf(lf[1], sizeof lf[1], "encoder.%s_decoder.linear.bias", nm[which]);
    const float* lw = W(w, lf[0], 0, 0, &err);
    const float* lb = W(w, lf[1], 0, 0, &err);
    if (err) return err;
    const long rows = (long)B * T, n = rows * dim;
    float* a = (float*)malloc(sizeof(float) * (size_t)n);
    float* b2 = (float*)malloc(sizeof(float) * (size_t)n);
    conv1d_cl(fused, B, T, dim, p[0], p[1], dim, 3, 1, 1, a);
    for (long i = 0; i < n; ++i) a[i] = a[i] > 0 ? a[i] : 0;
    layernorm(a, rows, dim, p[2], p[3]);
    for (long i = 0; i < n; ++i) a[i] = a[i] > 0 ? a[i] : 0;
    conv1d_cl(a, B, T, dim, p[4], p[5], dim, 3, 1, 1, b2);
    for (long i = 0; i < n; ++i) b2[i] = b2[i] > 0 ? b2[i] : 0;
    linear(b2, rows, dim, lw, lb, 1, pred);
    if (which == 2)
        for (long i = 0; i < rows; ++i) pred[i] = pred[i] > 0 ? pred[i] : 0;
    if (features) {
        layernorm(b2, rows, dim, p[6], p[7]);
        memcpy(features, b2, sizeof(float) * (size_t)n);
    }
    free(a);
    free(b2);
    return ESO_OK;
}

/* torch.bucketize(v, edges, right=False): number of edges strictly below v (networks.py:128-149) */
static int bucketize(float v, const float* edges, int n) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) / 2;
        if (edges[mid] < v) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

/* ---------------------------------------------------------------- PhonemeEncoder.forward (up to durations)
 * layers/networks.py:336-384.
 *   mask            uint8 (B,T) or NULL (B==1)
 *   pitch_t/energy_t float (B,T) teacher values or NULL (train=False -> predictions are bucketised)
 *   dur_t           int32 (B,T) forced durations or NULL (-> round(duration_pred), half-to-even)
 * outputs: pitch/energy/dur pred (B,T); pitch/energy bucket index int32 (B,T);
 *          feat (B,T,4*dim) = cat[fused, pitch_emb, energy_emb, duration_features] with padded rows 0;
 *          dur int32 (B,T) = clamp(masked_fill(round(dur_pred), mask, 0), 0); mel_len int32 (B).
 * Optional stage taps (may be NULL): f_taps[i] block outputs, fused_tap (B,T,dim).           */
int eso_phoneme_encoder(const eso_cfg* c, const eso_weights* w, int B, int T, const int32_t* phoneme,
                        const uint8_t* mask, const float* pitch_t, const float* energy_t, const int32_t* dur_t,
                        float* pitch_pred, float* energy_pred, float* dur_pred, int32_t* pitch_idx,
                        int32_t* energy_idx, float* feat, int32_t* dur, int32_t* mel_len, float* const* f_taps,
                        float* fused_tap) {
    int err = ESO_OK;
    const int dim = c->embed_dim / c->reduction, D = c->depth;
    if (B > 1 && !mask) return ESO_ERR_ARG; /* networks.py:338 raises KeyError */
    if (B == 1) mask = NULL;
    float** feats = (float**)calloc(D, sizeof(float*));
    uint8_t** bm = (uint8_t**)calloc(D, sizeof(uint8_t*));
    for (int i = 0; i < D; ++i) {
        const int n = eso_block_len(c, T, i);
        feats[i] = (float*)malloc(sizeof(float) * (size_t)B * n * (dim << i));
        bm[i] = (uint8_t*)malloc((size_t)B * n);
    }
    const long rows = (long)B * T;
    float* fused = (float*)malloc(sizeof(float) * (size_t)rows * dim);
    float* dfeat = (float*)malloc(sizeof(float) * (size_t)rows * dim);
    err = eso_encoder(c, w, B, T, phoneme, mask, feats, bm);
    const uint8_t* m0 = mask ? bm[0] : NULL; /* == phoneme_mask: block 0 has pool 1 */
    if (!err) err = eso_fuse(c, w, B, T, (const float* const*)feats, m0, fused);
    if (!err) err = eso_acoustic(c, w, 0, B, T, fused, pitch_pred, NULL);
    if (!err) err = eso_acoustic(c, w, 1, B, T, fused, energy_pred, NULL);
    if (!err) err = eso_acoustic(c, w, 2, B, T, fused, dur_pred, dfeat);
    const float* pbins = W(w, "encoder.pitch_decoder.pitch_bins", 0, 0, &err);
    const float* ebins = W(w, "encoder.energy_decoder.energy_bins", 0, 0, &err);
    const float* pemb = W(w, "encoder.pitch_decoder.pitch_embedding.weight", 0, 0, &err);
    const float* eemb = W(w, "encoder.energy_decoder.energy_embedding.weight", 0, 0, &err);
    if (!err) {
        for (long r = 0; r < rows; ++r) {
            const int pad = m0 ? m0[r] : 0;
            const int pi = bucketize(pitch_t ? pitch_t[r] : pitch_pred[r], pbins, dim - 1);   /* :128-139 */
            const int ei = bucketize(energy_t ? energy_t[r] : energy_pred[r], ebins, dim - 1); /* :141-149 */
            pitch_idx[r] = pi;
            energy_idx[r] = ei;
            float* fr = feat + r * 4 * dim; /* :370-371 cat order */
            for (int cc = 0; cc < dim; ++cc) {
                fr[cc] = fused[r * dim + cc]; /* already masked by Fuse */
                fr[dim + cc] = pad ? 0.f : pemb[(long)pi * dim + cc];       /* :352-353 */
                fr[2 * dim + cc] = pad ? 0.f : eemb[(long)ei * dim + cc];   /* :359-361 */
                fr[3 * dim + cc] = pad ? 0.f : dfeat[r * dim + cc];         /* :366-368 */
            }
            float d = dur_t ? (float)dur_t[r] : round_half_even(dur_pred[r]); /* :379-380 */
            if (mask) {                                                      /* :381-382 */
                if (mask[r]) d = 0.f;
                if (d < 0.f) d = 0.f;
            }
            dur[r] = (int32_t)d; /* FeatureUpsampler `.int()`, :234 */
        }
        for (int b = 0; b < B; ++b) {
            long s = 0;
            for (int t = 0; t < T; ++t) s += dur[b * T + t] > 0 ? dur[b * T + t] : 0;
            mel_len[b] = (int32_t)s; /* :237 */
        }
        if (f_taps)
            for (int i = 0; i < D; ++i)
                if (f_taps[i]) memcpy(f_taps[i], feats[i], sizeof(float) * (size_t)B * eso_block_len(c, T, i) * (dim << i));
        if (fused_tap) memcpy(fused_tap, fused, sizeof(float) * (size_t)rows * dim);
    }
    for (int i = 0; i < D; ++i) { free(feats[i]); free(bm[i]); }
    free(feats); free(bm); free(fused); free(dfeat);
    return err;
}

/* ---------------------------------------------------------------- FeatureUpsampler = length regulator
 * layers/networks.py:228-258 (live) and layers/acoustic.py:33-42 (dead twin, same rule):
 * frame j of utterance b comes from phoneme i with cumsum(d)[i-1] <= j < cumsum(d)[i];
 * rows beyond mel_len[b] are padding (feature 0.0, mask True).  idx[b][j] = i or -1.
 * negative repeats: repeat_interleave raises on negatives; durations are clamped >= 0 upstream
 * for B>1 and are ReLU'd+rounded for B==1, so max(d,0) (acoustic.py:39) is the shared rule. */
void eso_length_regulate(int B, int T, const int32_t* dur, int L, int32_t* idx) {
    for (int b = 0; b < B; ++b) {
        int j = 0;
        for (int t = 0; t < T; ++t) {
            const int d = dur[b * T + t] > 0 ? dur[b * T + t] : 0;
            for (int r = 0; r < d; ++r, ++j)
                if (j < L) idx[(long)b * L + j] = t;
        }
        for (; j < L; ++j) idx[(long)b * L + j] = -1;
    }
}

/* features (B,L,C) / masks uint8 (B,L) from the (B,T,C) phoneme-rate tensor and idx.
 * fmask: the (B,T) phoneme padding mask or NULL; padded output frames get mask 1 (:246-249). */
void eso_upsample(int B, int T, int C, int L, const float* feat, const uint8_t* fmask, const int32_t* idx, float* out,
                  uint8_t* omask) {
    for (int b = 0; b < B; ++b)
        for (int j = 0; j < L; ++j) {
            const int i = idx[(long)b * L + j];
            float* o = out + ((long)b * L + j) * C;
            if (i < 0) {
                memset(o, 0, sizeof(float) * C);
                if (omask) omask[(long)b * L + j] = 1;
            } else {
                memcpy(o, feat + ((long)b * T + i) * C, sizeof(float) * C);
                if (omask) omask[(long)b * L + j] = fmask ? fmask[b * T + i] : 0;
            }
        }
}

/* ---------------------------------------------------------------- MelDecoder.forward
 * layers/networks.py:291-304.  skip = LN(tanh(Linear(d4,dx2)(x)));  per block: x = skip;
 * depth x [x = LN(tanh(Conv1x1(dwConv_k(x))))]; skip = LN_skip(x + skip);  mel = Linear(dx2,80)(skip).
 * dwConv zero-pads at the ends of the PADDED length L, so padded frames are computed and leak
 * into the last valid frames exactly as in the reference.                                     */
int eso_mel_decoder(const eso_cfg* c, const eso_weights* w, int B, int L, const float* features, float* mel) {
    int err = ESO_OK;
    const int dim = c->embed_dim / c->reduction, d4 = 4 * dim, dx2 = d4 < 256 ? d4 : 256, k = c->dec_kernel;
    const long rows = (long)B * L;
    const float* pw = W(w, "decoder.proj.0.weight", 0, 0, &err);
    const float* pb = W(w, "decoder.proj.0.bias", 0, 0, &err);
    const float* pg = W(w, "decoder.proj.2.weight", 0, 0, &err);
    const float* pbb = W(w, "decoder.proj.2.bias", 0, 0, &err);
    const float* mw = W(w, "decoder.mel_linear.weight", 0, 0, &err);
    const float* mb = W(w, "decoder.mel_linear.bias", 0, 0, &err);
    if (err) return err;
    float* skip = (float*)malloc(sizeof(float) * (size_t)rows * dx2);
    float* x = (float*)malloc(sizeof(float) * (size_t)rows * dx2);
    float* t = (float*)malloc(sizeof(float) * (size_t)rows * dx2);
    linear(features, rows, d4, pw, pb, dx2, skip);
#pragma omp parallel for schedule(static)
    for (long i = 0; i < rows * dx2; ++i) skip[i] = (float)tanh((double)skip[i]);
    layernorm(skip, rows, dx2, pg, pbb);
    for (int b = 0; b < c->n_blocks && !err; ++b) {
        memcpy(x, skip, sizeof(float) * (size_t)rows * dx2);
        for (int d = 0; d < c->block_depth && !err; ++d) {
            const float* dw = W(w, "decoder.blocks.%d.0.%d.0.0.weight", b, d, &err);
            const float* db = W(w, "decoder.blocks.%d.0.%d.0.0.bias", b, d, &err);
            const float* qw = W(w, "decoder.blocks.%d.0.%d.0.1.weight", b, d, &err);
            const float* qb = W(w, "decoder.blocks.%d.0.%d.0.1.bias", b, d, &err);
            const float* lg = W(w, "decoder.blocks.%d.0.%d.1.weight", b, d, &err);
            const float* lb = W(w, "decoder.blocks.%d.0.%d.1.bias", b, d, &err);
            if (err) break;
            dwconv1d_cl(x, B, L, dx2, dw, db, k, t);
            linear(t, rows, dx2, qw, qb, dx2, x);
#pragma omp parallel for schedule(static)
            for (long i = 0; i < rows * dx2; ++i) x[i] = (float)tanh((double)x[i]);
            layernorm(x, rows, dx2, lg, lb);
        }
        const float* sg = W(w, "decoder.blocks.%d.1.weight", b, 0, &err);
        const float* sb = W(w, "decoder.blocks.%d.1.bias", b, 0, &err);
        if (err) break;
        add_inplace(skip, x, rows * dx2); /* :299 skip_norm(x + skip) */
        layernorm(skip, rows, dx2, sg, sb);
    }
    if (!err) linear(skip, rows, dx2, mw, mb, c->n_mel, mel);
    free(skip); free(x); free(t);
    return err;
}

/* Phoneme2Mel.forward tail, layers/networks.py:424-427: mel.masked_fill(mask[:, :, :80], 0)
 * when masks is not None and B > 1.                                                          */
void eso_mask_mel(int B, int L, int n_mel, const uint8_t* omask, float* mel) {
    if (!omask || B <= 1) return;
    mask_rows(mel, omask, (long)B * L, n_mel);
}

/* utils/tools.py:43-51 get_mask_from_lengths: mask[b][t] = t >= lengths[b] */
/* ---------------------------------------------------------------- HiFi-GAN generator
 * hifigan/models.py:84-135 (Generator.forward) with ResBlock1 (:20-58) / ResBlock2 (:61-82), weights as after
 * remove_weight_norm() (model.py:44: `weight`, `bias` per conv).  Channels-last restatement: x (B, L, C).
 *   x = conv_pre(mel); for each upsample stage i: x = ConvTranspose1d_i(leaky_relu(x, 0.1));
 *       x = mean_j ResBlock_{i,j}(x);   x = tanh(conv_post(leaky_relu(x)))   [the last leaky_relu uses the default slope 0.01]
 * Conv1d: weight (Cout,Cin,k), dilation d, padding (k*d - d)/2 (get_padding, :14-15).
 * ConvTranspose1d: weight (Cin,Cout,k), stride u, padding (k-u)//2 -> length L*u.                                        */
typedef struct {
    int n_mel, initial_channel, n_up, n_kernels, resblock; /* resblock: 1 or 2 */
    int up_rates[8], up_kernels[8];
    int rb_kernels[8], rb_dilations[8][3];
} eso_hifigan_cfg;

static void lrelu_copy(const float* x, long n, float slope, float* y) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i) y[i] = x[i] > 0 ? x[i] : x[i] * slope;
}

static void conv1d_dil_cl(const float* x, int B, int N, int Cin, const float* Wt, const float* bias, int Cout, int k, int dil,
                          float* y) {
    const int pad = (k * dil - dil) / 2;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < N; ++t) {
            float* yr = y + ((long)b * N + t) * Cout;
            for (int co = 0; co < Cout; ++co) {
                acc_t s = (acc_t)bias[co];
                for (int j = 0; j < k; ++j) {
                    const int ti = t + j * dil - pad;
                    if (ti < 0 || ti >= N) continue;
                    const float* xr = x + ((long)b * N + ti) * Cin;
                    const float* wr = Wt + (long)co * Cin * k + j;
                    for (int ci = 0; ci < Cin; ++ci) s += (acc_t)xr[ci] * (acc_t)wr[(long)ci * k];
                }
                yr[co] = (float)s;
            }
        }
}

/* y[b][n*u + j - pad][co] += x[b][n][ci] * W[ci][co][j], gathered per output position */
static void convT1d_cl(const float* x, int B, int N, int Cin, const float* Wt, const float* bias, int Cout, int k, int u, float* y) {
    const int pad = (k - u) / 2, Nout = (N - 1) * u - 2 * pad + k;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < Nout; ++t) {
            float* yr = y + ((long)b * Nout + t) * Cout;
            for (int co = 0; co < Cout; ++co) {
                acc_t s = (acc_t)bias[co];
                for (int j = 0; j < k; ++j) {
                    const int q = t + pad - j;
                    if (q < 0 || q % u) continue;
                    const int n = q / u;
                    if (n >= N) continue;
                    const float* xr = x + ((long)b * N + n) * Cin;
                    for (int ci = 0; ci < Cin; ++ci) s += (acc_t)xr[ci] * (acc_t)Wt[((long)ci * Cout + co) * k + j];
                }
                yr[co] = (float)s;
            }
        }
}

int eso_hifigan(const eso_hifigan_cfg* c, const eso_weights* w, int B, int L, const float* mel, float* wav) {
    int err = ESO_OK;
    if (c->n_up < 1 || c->n_up > 8 || c->n_kernels < 1 || c->n_kernels > 8 || B < 1 || L < 1) return ESO_ERR_ARG;
    long N = L;
    int C = c->initial_channel;
    float* x = (float*)malloc(sizeof(float) * (size_t)B * N * C);
    conv1d_dil_cl(mel, B, L, c->n_mel, W(w, "conv_pre.weight", 0, 0, &err), W(w, "conv_pre.bias", 0, 0, &err), C, 7, 1, x);
    if (err) { free(x); return err; }
    for (int i = 0; i < c->n_up; ++i) {
        const int u = c->up_rates[i], k = c->up_kernels[i], Co = C / 2;
        const long No = N * u;
        float* a = (float*)malloc(sizeof(float) * (size_t)B * N * C);
        lrelu_copy(x, (long)B * N * C, 0.1f, a);
        float* y = (float*)malloc(sizeof(float) * (size_t)B * No * Co);
        convT1d_cl(a, B, (int)N, C, W(w, "ups.%d.weight", i, 0, &err), W(w, "ups.%d.bias", i, 0, &err), Co, k, u, y);
        free(a); free(x);
        if (err) { free(y); return err; }
        N = No; C = Co;
        const long n = (long)B * N * C;
        float* xs = (float*)calloc((size_t)n, sizeof(float));
        float* r = (float*)malloc(sizeof(float) * (size_t)n);
        float* t1 = (float*)malloc(sizeof(float) * (size_t)n);
        float* t2 = (float*)malloc(sizeof(float) * (size_t)n);
        for (int j = 0; j < c->n_kernels; ++j) {
            const int rb = i * c->n_kernels + j, kk = c->rb_kernels[j];
            memcpy(r, y, sizeof(float) * (size_t)n);
            const int nconv = c->resblock == 1 ? 3 : 2;
            for (int m = 0; m < nconv; ++m) {
                char f1[96], f2[96];
                lrelu_copy(r, n, 0.1f, t1);
                if (c->resblock == 1) {
                    snprintf(f1, sizeof f1, "resblocks.%d.convs1.%d.weight", rb, m);
                    snprintf(f2, sizeof f2, "resblocks.%d.convs1.%d.bias", rb, m);
                    conv1d_dil_cl(t1, B, (int)N, C, W(w, f1, 0, 0, &err), W(w, f2, 0, 0, &err), C, kk, c->rb_dilations[j][m], t2);
                    lrelu_copy(t2, n, 0.1f, t1);
                    snprintf(f1, sizeof f1, "resblocks.%d.convs2.%d.weight", rb, m);
                    snprintf(f2, sizeof f2, "resblocks.%d.convs2.%d.bias", rb, m);
                    conv1d_dil_cl(t1, B, (int)N, C, W(w, f1, 0, 0, &err), W(w, f2, 0, 0, &err), C, kk, 1, t2);
                } else {
                    snprintf(f1, sizeof f1, "resblocks.%d.convs.%d.weight", rb, m);
                    snprintf(f2, sizeof f2, "resblocks.%d.convs.%d.bias", rb, m);
                    conv1d_dil_cl(t1, B, (int)N, C, W(w, f1, 0, 0, &err), W(w, f2, 0, 0, &err), C, kk, c->rb_dilations[j][m], t2);
                }
                if (err) break;
                for (long e = 0; e < n; ++e) r[e] = t2[e] + r[e];          /* x = xt + x */
            }
            for (long e = 0; e < n; ++e) xs[e] += r[e];
        }
        for (long e = 0; e < n; ++e) xs[e] /= (float)c->n_kernels;       /* x = xs / num_kernels */
        free(r); free(t1); free(t2); free(y);
        x = xs;
        if (err) { free(x); return err; }
    }
    {
        const long n = (long)B * N * C;
        float* a = (float*)malloc(sizeof(float) * (size_t)n);
        lrelu_copy(x, n, 0.01f, a);                                        /* F.leaky_relu(x): default slope */
        conv1d_dil_cl(a, B, (int)N, C, W(w, "conv_post.weight", 0, 0, &err), W(w, "conv_post.bias", 0, 0, &err), 1, 7, 1, wav);
        free(a); free(x);
        if (err) return err;
        for (long e = 0; e < (long)B * N; ++e) wav[e] = (float)tanh((double)wav[e]);
    }
    return ESO_OK;
}

void eso_mask_from_lengths(int B, int T, const int32_t* lengths, uint8_t* mask) {
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < T; ++t) mask[b * T + t] = (uint8_t)(t >= lengths[b]);
}

/*
 * oracle/vit_oracle.c  --  TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the staghado/vit.cpp forward path (vit_model_load ->
 * vit_image_preprocess -> vit_encode_image/vit_predict) with ggml's CPU-backend
 * rounding points (fp16-rounded mul_mat activations, f32 accumulate, fp16 LUT for
 * exp and tanh-GELU, double-sum LayerNorm/softmax).  It exists so that the HIP
 * engine in vit.cpp_amd/csrc has something to be compared with: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library,
 * and only as the checker.  The product never links or calls it.
 *
 * PARITY UNPINNED: the reference cannot be built here (its ggml submodule is an
 * empty directory, /root/reference/ggml, and it has no unit tests or golden
 * vectors -- SURVEY.md 8c).  ggml (ggerganov/ggml, git submodule, pinned SHA not
 * recoverable; API window ~2023-11-13..2023-12-21) holds the arithmetic; its
 * published CPU algorithm for each op is restated below and anchored on the
 * reference's own call sites (vit.cpp line numbers cited per function).  The only
 * independent cross-check available offline is HuggingFace transformers' ViT in
 * f32 (tests/test_cpu_oracle.py::test_oracle_vs_transformers_vit_f32), which pins the architecture but not
 * ggml's rounding points.
 *
 * Build: gcc -O3 -mavx2 -mfma -mf16c -ffp-contract=off -fopenmp -shared -fPIC
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <immintrin.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ fp16/bf16 */
/* ggml on x86+F16C: GGML_FP32_TO_FP16 = _cvtss_sh(x,0) (round-nearest-even). */
static inline uint16_t f32_to_f16(float x) { return (uint16_t)_cvtss_sh(x, 0); }
static inline float f16_to_f32(uint16_t h) { return _cvtsh_ss(h); }
static inline float round_f16(float x) { return f16_to_f32(f32_to_f16(x)); }
static inline float round_bf16(float x) {
    uint32_t u; memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return x;           /* NaN */
    u += 0x7fffu + ((u >> 16) & 1u); u &= 0xffff0000u;
    float r; memcpy(&r, &u, 4); return r;
}
/* rounding selector: 0 none, 1 fp16, 2 bf16 */
static inline float round_sel(float x, int sel) {
    return sel == 1 ? round_f16(x) : sel == 2 ? round_bf16(x) : x;
}

/* --------------------------------------------------------------------- tables */
/* ggml_init(): table_exp_f16[i] = FP32_TO_FP16(expf(FP16_TO_FP32(i))),
 *              table_gelu_f16[i] = FP32_TO_FP16(ggml_gelu_f32(f)) with the tanh form
 *              0.5f*x*(1.0f + tanhf(SQRT_2_OVER_PI*x*(1.0f + GELU_COEF_A*x*x))). */
static uint16_t g_table_exp[65536];
static uint16_t g_table_gelu[65536];
static int g_tables_ready = 0;
static __thread int g_dot_exact = 0;
static const float GELU_COEF_A = 0.044715f;
static const float SQRT_2_OVER_PI = 0.79788456080286535587989211986876f;
static inline float gelu_f32(float x) {
    return 0.5f * x * (1.0f + tanhf(SQRT_2_OVER_PI * x * (1.0f + GELU_COEF_A * x * x)));
}
static void init_tables(void) {
    if (g_tables_ready) return;
    for (int i = 0; i < 65536; ++i) {
        float f = f16_to_f32((uint16_t)i);
        g_table_exp[i] = f32_to_f16(expf(f));
        g_table_gelu[i] = f32_to_f16(gelu_f32(f));
    }
    g_tables_ready = 1;
}

/* ------------------------------------------------------------------- dot core */
/* ggml_vec_dot_f32 / ggml_vec_dot_f16 (AVX2: 4 x 8-lane f32 FMA accumulators,
 * GGML_F32_STEP = GGML_F16_STEP = 32), reduction r0+=r2, r1+=r3, r0+=r1, then the
 * 8 lanes; leftovers added in scalar.  Summation order is implementation-defined
 * in ggml; this mirrors the AVX2 build. */
static inline float dot_f32(const float *a, const float *b, int n) {
    if (g_dot_exact) {      /* probe: double accumulation (a float x float product is exact in double); 4 x 4 lanes, then the leftovers */
        __m256d s0 = _mm256_setzero_pd(), s1 = s0, s2 = s0, s3 = s0;
        int i = 0;
        for (; i + 16 <= n; i += 16) {
            s0 = _mm256_fmadd_pd(_mm256_cvtps_pd(_mm_loadu_ps(a + i)), _mm256_cvtps_pd(_mm_loadu_ps(b + i)), s0);
            s1 = _mm256_fmadd_pd(_mm256_cvtps_pd(_mm_loadu_ps(a + i + 4)), _mm256_cvtps_pd(_mm_loadu_ps(b + i + 4)), s1);
            s2 = _mm256_fmadd_pd(_mm256_cvtps_pd(_mm_loadu_ps(a + i + 8)), _mm256_cvtps_pd(_mm_loadu_ps(b + i + 8)), s2);
            s3 = _mm256_fmadd_pd(_mm256_cvtps_pd(_mm_loadu_ps(a + i + 12)), _mm256_cvtps_pd(_mm_loadu_ps(b + i + 12)), s3);
        }
        double t[4]; _mm256_storeu_pd(t, _mm256_add_pd(_mm256_add_pd(s0, s1), _mm256_add_pd(s2, s3)));
        double s = (t[0] + t[1]) + (t[2] + t[3]);
        for (; i < n; ++i) s += (double)a[i] * (double)b[i];
        return (float)s;
    }
    float acc[32];
    for (int j = 0; j < 32; ++j) acc[j] = 0.0f;
    const int np = n & ~31;
    for (int i = 0; i < np; i += 32)
        for (int j = 0; j < 32; ++j) acc[j] = __builtin_fmaf(a[i + j], b[i + j], acc[j]);
    for (int j = 0; j < 8; ++j) { acc[j] += acc[16 + j]; acc[8 + j] += acc[24 + j]; }
    for (int j = 0; j < 8; ++j) acc[j] += acc[8 + j];
    float t0 = acc[0] + acc[4], t1 = acc[1] + acc[5], t2 = acc[2] + acc[6], t3 = acc[3] + acc[7];
    float sumf = (t0 + t1) + (t2 + t3);
    for (int i = np; i < n; ++i) sumf += a[i] * b[i];
    return sumf;
}

/* ------------------------------------------------------------ quantised blocks */
#define QK 32
enum { T_F32 = 0, T_F16 = 1, T_Q4_0 = 2, T_Q4_1 = 3, T_Q5_0 = 6, T_Q5_1 = 7, T_Q8_0 = 8 };
static int type_block_bytes(int t) {
    switch (t) { case T_F32: return 4; case T_F16: return 2; case T_Q4_0: return 18; case T_Q4_1: return 20;
                 case T_Q5_0: return 22; case T_Q5_1: return 24; case T_Q8_0: return 34; }
    return 0;
}
static int type_block_elems(int t) { return (t == T_F32 || t == T_F16) ? 1 : QK; }

typedef struct { float d; float s; int8_t qs[QK]; } blk_q8;   /* q8_0 (d is fp16-rounded) / q8_1 (d,s f32) */

/* quantize_row_q8_0_reference: d = amax/127, q = roundf(x/d); d stored as fp16. */
static void quant_q8_0(const float *x, blk_q8 *y, int n) {
    for (int b = 0; b < n / QK; ++b) {
        float amax = 0.0f;
        for (int j = 0; j < QK; ++j) { float v = fabsf(x[b * QK + j]); if (v > amax) amax = v; }
        const float d = amax / 127.0f; const float id = d ? 1.0f / d : 0.0f;
        y[b].d = round_f16(d); y[b].s = 0.0f;
        for (int j = 0; j < QK; ++j) y[b].qs[j] = (int8_t)roundf(x[b * QK + j] * id);
    }
}
/* quantize_row_q8_1_reference: as q8_0 but d kept in f32 and s = d*sum(q). */
static void quant_q8_1(const float *x, blk_q8 *y, int n) {
    for (int b = 0; b < n / QK; ++b) {
        float amax = 0.0f;
        for (int j = 0; j < QK; ++j) { float v = fabsf(x[b * QK + j]); if (v > amax) amax = v; }
        const float d = amax / 127.0f; const float id = d ? 1.0f / d : 0.0f;
        int sum = 0;
        for (int j = 0; j < QK; ++j) { int8_t q = (int8_t)roundf(x[b * QK + j] * id); y[b].qs[j] = q; sum += q; }
        y[b].d = d; y[b].s = sum * d;
    }
}
static inline float rd_f16(const uint8_t *p) { uint16_t h; memcpy(&h, p, 2); return f16_to_f32(h); }

/* ggml_vec_dot_q{4_0,5_0,8_0}_q8_0 / q{4_1,5_1}_q8_1, scalar form. */
static float dot_quant(int type, const uint8_t *w, const blk_q8 *y, int n) {
    float sumf = 0.0f;
    const int nb = n / QK, bs = type_block_bytes(type);
    for (int b = 0; b < nb; ++b) {
        const uint8_t *p = w + (size_t)b * bs;
        int sumi = 0;
        switch (type) {
        case T_Q4_0: { const uint8_t *qs = p + 2;
            for (int j = 0; j < 16; ++j) sumi += ((qs[j] & 0x0F) - 8) * y[b].qs[j] + ((qs[j] >> 4) - 8) * y[b].qs[j + 16];
            sumf += sumi * rd_f16(p) * y[b].d; } break;
        case T_Q4_1: { const uint8_t *qs = p + 4;
            for (int j = 0; j < 16; ++j) sumi += (qs[j] & 0x0F) * y[b].qs[j] + (qs[j] >> 4) * y[b].qs[j + 16];
            sumf += (rd_f16(p) * y[b].d) * sumi + rd_f16(p + 2) * y[b].s; } break;
        case T_Q5_0: { uint32_t qh; memcpy(&qh, p + 2, 4); const uint8_t *qs = p + 6;
            for (int j = 0; j < 16; ++j) {
                const uint8_t xh0 = ((qh >> (j + 0)) << 4) & 0x10, xh1 = (qh >> (j + 12)) & 0x10;
                sumi += (((qs[j] & 0x0F) | xh0) - 16) * y[b].qs[j] + (((qs[j] >> 4) | xh1) - 16) * y[b].qs[j + 16]; }
            sumf += (rd_f16(p) * y[b].d) * sumi; } break;
        case T_Q5_1: { uint32_t qh; memcpy(&qh, p + 4, 4); const uint8_t *qs = p + 8;
            for (int j = 0; j < 16; ++j) {
                const uint8_t xh0 = ((qh >> (j + 0)) << 4) & 0x10, xh1 = (qh >> (j + 12)) & 0x10;
                sumi += ((qs[j] & 0x0F) | xh0) * y[b].qs[j] + ((qs[j] >> 4) | xh1) * y[b].qs[j + 16]; }
            sumf += (rd_f16(p) * y[b].d) * sumi + rd_f16(p + 2) * y[b].s; } break;
        case T_Q8_0: { const int8_t *qs = (const int8_t *)(p + 2);
            for (int j = 0; j < QK; ++j) sumi += qs[j] * y[b].qs[j];
            sumf += sumi * (rd_f16(p) * y[b].d); } break;
        }
    }
    return sumf;
}
/* dequantize_row_q*: used to export weights to the GPU tests and for mode 'ideal'. */
static void dequant_row(int type, const uint8_t *w, float *out, int n) {
    const int nb = n / QK, bs = type_block_bytes(type);
    for (int b = 0; b < nb; ++b) {
        const uint8_t *p = w + (size_t)b * bs; float *o = out + b * QK;
        switch (type) {
        case T_Q4_0: { float d = rd_f16(p); const uint8_t *qs = p + 2;
            for (int j = 0; j < 16; ++j) { o[j] = ((qs[j] & 0x0F) - 8) * d; o[j + 16] = ((qs[j] >> 4) - 8) * d; } } break;
        case T_Q4_1: { float d = rd_f16(p), m = rd_f16(p + 2); const uint8_t *qs = p + 4;
            for (int j = 0; j < 16; ++j) { o[j] = (qs[j] & 0x0F) * d + m; o[j + 16] = (qs[j] >> 4) * d + m; } } break;
        case T_Q5_0: { float d = rd_f16(p); uint32_t qh; memcpy(&qh, p + 2, 4); const uint8_t *qs = p + 6;
            for (int j = 0; j < 16; ++j) { const uint8_t xh0 = ((qh >> (j + 0)) << 4) & 0x10, xh1 = (qh >> (j + 12)) & 0x10;
                o[j] = (((qs[j] & 0x0F) | xh0) - 16) * d; o[j + 16] = (((qs[j] >> 4) | xh1) - 16) * d; } } break;
        case T_Q5_1: { float d = rd_f16(p), m = rd_f16(p + 2); uint32_t qh; memcpy(&qh, p + 4, 4); const uint8_t *qs = p + 8;
            for (int j = 0; j < 16; ++j) { const uint8_t xh0 = ((qh >> (j + 0)) << 4) & 0x10, xh1 = (qh >> (j + 12)) & 0x10;
                o[j] = ((qs[j] & 0x0F) | xh0) * d + m; o[j + 16] = ((qs[j] >> 4) | xh1) * d + m; } } break;
        case T_Q8_0: { float d = rd_f16(p); const int8_t *qs = (const int8_t *)(p + 2);
            for (int j = 0; j < QK; ++j) o[j] = qs[j] * d; } break;
        }
    }
}

/* --------------------------------------------------------------------- tensors */
typedef struct {
    int type;            /* T_* as stored in the file */
    int n_dims;
    int64_t ne[4];
    size_t nbytes;
    uint8_t *raw;        /* file bytes */
    float *f32;          /* expansion to f32 (exact for f32/f16; dequantised for q*) */
} otensor;

typedef struct {
    int D, L, H, C, P, S, ftype;
    int Cin, R;              /* input channels (3; 1 = ViTSTR file, extensions/vitstr.cpp/vitstr.cpp:482) and head rows per image (1; 25 = ViTSTR, :864-883) */
    int g, N;            /* patches per side, tokens */
    float eps;
    int n_labels; int *label_keys; char **labels;
    otensor cls, pos, pe_w, pe_b;
    otensor *ln1_w, *ln1_b, *qkv_w, *qkv_b, *proj_w, *proj_b, *ln2_w, *ln2_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
    otensor norm_w, norm_b, head_w, head_b;
    int n_tensors_loaded;
} omodel;

typedef struct {
    int act_round;   /* rounding of mul_mat activations: 0 none, 1 fp16 (ggml with f16 weights), 2 bf16 */
    int lut;         /* exp/GELU: 0 plain f32 expf/tanhf, 1 ggml fp16 LUT (in+out rounded to fp16), 2 = the bf16 engine: the class softmax
                        rounds in+out to bf16; GELU and the attention softmax round the OUTPUT only (r03: the reference has no bf16
                        rounding points to reproduce, so the perf mode does not pay for a rounded exponent / GELU argument) */
    int attn_round;  /* q,k,v (and p) rounding before the attention products: 0 f32 (ggml), 1 fp16, 2 bf16 */
    int w_round;     /* extra rounding of the (already f16/f32) dense weights: 0 none, 2 bf16 */
    int quant_act;   /* 1: with q* weights quantise activations to q8_0/q8_1 like ggml; 0: dequantised weights x act_round; 2 (probe): as 1, from the act_round-rounded activation */
    int dot_exact;   /* probe: 1 = accumulate every dot product in double (measures ggml's own summation-order noise floor) */
    int attn_qk_round, attn_v_round; /* probe: override attn_round separately for q,k and for v (-1 = follow attn_round) */
} omode;


static void tensor_free(otensor *t) { free(t->raw); free(t->f32); t->raw = NULL; t->f32 = NULL; }

/* ------------------------------------------------------------------ file loader */
/* Follows vit.cpp:308-712 (reader) and convert-pth-to-ggml.py:105-158 (writer). */
static int read_i32(FILE *f, int32_t *v) { return fread(v, 4, 1, f) == 1; }

static otensor *model_slot(omodel *m, const char *name) {
    if (!strcmp(name, "cls_token")) return &m->cls;
    if (!strcmp(name, "pos_embed")) return &m->pos;
    if (!strcmp(name, "patch_embed.proj.weight")) return &m->pe_w;
    if (!strcmp(name, "patch_embed.proj.bias")) return &m->pe_b;
    if (!strcmp(name, "norm.weight")) return &m->norm_w;
    if (!strcmp(name, "norm.bias")) return &m->norm_b;
    if (!strcmp(name, "head.weight")) return &m->head_w;
    if (!strcmp(name, "head.bias")) return &m->head_b;
    int i = -1; char rest[64];
    if (sscanf(name, "blocks.%d.%63s", &i, rest) == 2 && i >= 0 && i < m->L) {
        if (!strcmp(rest, "norm1.weight")) return &m->ln1_w[i];
        if (!strcmp(rest, "norm1.bias")) return &m->ln1_b[i];
        if (!strcmp(rest, "attn.qkv.weight")) return &m->qkv_w[i];
        if (!strcmp(rest, "attn.qkv.bias")) return &m->qkv_b[i];
        if (!strcmp(rest, "attn.proj.weight")) return &m->proj_w[i];
        if (!strcmp(rest, "attn.proj.bias")) return &m->proj_b[i];
        if (!strcmp(rest, "norm2.weight")) return &m->ln2_w[i];
        if (!strcmp(rest, "norm2.bias")) return &m->ln2_b[i];
        if (!strcmp(rest, "mlp.fc1.weight")) return &m->fc1_w[i];
        if (!strcmp(rest, "mlp.fc1.bias")) return &m->fc1_b[i];
        if (!strcmp(rest, "mlp.fc2.weight")) return &m->fc2_w[i];
        if (!strcmp(rest, "mlp.fc2.bias")) return &m->fc2_b[i];
    }
    return NULL;
}

void oracle_model_free(omodel *m);

omodel *oracle_model_load(const char *path) {
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "oracle: cannot open %s\n", path); return NULL; }
    omodel *m = (omodel *)calloc(1, sizeof(omodel));
    int32_t magic = 0;
    if (!read_i32(f, &magic) || (uint32_t)magic != 0x67676d6cu) { fprintf(stderr, "oracle: bad magic\n"); fclose(f); free(m); return NULL; }
    int32_t hp[7];
    for (int i = 0; i < 7; ++i) if (!read_i32(f, &hp[i])) { fclose(f); free(m); return NULL; }
    m->D = hp[0]; m->L = hp[1]; m->H = hp[2]; m->C = hp[3]; m->P = hp[4]; m->S = hp[5]; m->ftype = hp[6] % 1000;
    m->g = m->S / m->P; m->N = m->g * m->g + 1; m->eps = 1e-6f;    /* vit.h:29, never read from the file */
    int32_t nl = 0; read_i32(f, &nl);
    m->n_labels = nl; m->label_keys = (int *)calloc(nl > 0 ? nl : 1, sizeof(int)); m->labels = (char **)calloc(nl > 0 ? nl : 1, sizeof(char *));
    for (int i = 0; i < nl; ++i) {
        int32_t key, len; read_i32(f, &key); read_i32(f, &len);
        m->label_keys[i] = key; m->labels[i] = (char *)calloc(len + 1, 1);
        if (len && fread(m->labels[i], 1, len, f) != (size_t)len) { fclose(f); oracle_model_free(m); return NULL; }
    }
    const int L = m->L;
    otensor **arrs[] = { &m->ln1_w, &m->ln1_b, &m->qkv_w, &m->qkv_b, &m->proj_w, &m->proj_b, &m->ln2_w, &m->ln2_b, &m->fc1_w, &m->fc1_b, &m->fc2_w, &m->fc2_b };
    for (int i = 0; i < 12; ++i) *arrs[i] = (otensor *)calloc(L, sizeof(otensor));
    for (;;) {
        int32_t n_dims, name_len, ttype;
        if (!read_i32(f, &n_dims)) break;
        if (!read_i32(f, &name_len) || !read_i32(f, &ttype)) break;
        int64_t ne[4] = {1, 1, 1, 1}, nel = 1;
        for (int i = 0; i < n_dims; ++i) { int32_t v; read_i32(f, &v); ne[i] = v; nel *= v; }
        char name[256] = {0};
        if (name_len > 255 || fread(name, 1, name_len, f) != (size_t)name_len) { oracle_model_free(m); fclose(f); return NULL; }
        otensor *t = model_slot(m, name);
        if (!t) { fprintf(stderr, "oracle: unknown tensor '%s'\n", name); oracle_model_free(m); fclose(f); return NULL; }
        const int bb = type_block_bytes(ttype), be = type_block_elems(ttype);
        if (!bb || (ne[0] % be)) { fprintf(stderr, "oracle: bad type %d for %s\n", ttype, name); oracle_model_free(m); fclose(f); return NULL; }
        t->type = ttype; t->n_dims = n_dims; memcpy(t->ne, ne, sizeof(ne));
        t->nbytes = (size_t)(nel / be) * bb;
        t->raw = (uint8_t *)malloc(t->nbytes);
        if (fread(t->raw, 1, t->nbytes, f) != t->nbytes) { fprintf(stderr, "oracle: short read %s\n", name); oracle_model_free(m); fclose(f); return NULL; }
        t->f32 = (float *)malloc((size_t)nel * 4);
        if (ttype == T_F32) memcpy(t->f32, t->raw, (size_t)nel * 4);
        else if (ttype == T_F16) { const uint16_t *h = (const uint16_t *)t->raw; for (int64_t i = 0; i < nel; ++i) t->f32[i] = f16_to_f32(h[i]); }
        else { const int64_t rows = nel / ne[0]; for (int64_t r = 0; r < rows; ++r) dequant_row(ttype, t->raw + (size_t)r * (ne[0] / QK) * bb, t->f32 + r * ne[0], (int)ne[0]); }
        m->n_tensors_loaded++;
    }
    fclose(f);
    m->Cin = (m->pe_w.n_dims == 4 && m->pe_w.ne[2] == 1) ? 1 : 3;
    m->R = m->Cin == 1 ? 25 : 1;
    if (m->n_tensors_loaded != 8 + 12 * L) { fprintf(stderr, "oracle: %d tensors, expected %d\n", m->n_tensors_loaded, 8 + 12 * L); oracle_model_free(m); return NULL; }
    init_tables();
    return m;
}

void oracle_model_free(omodel *m) {
    if (!m) return;
    otensor *single[] = { &m->cls, &m->pos, &m->pe_w, &m->pe_b, &m->norm_w, &m->norm_b, &m->head_w, &m->head_b };
    for (int i = 0; i < 8; ++i) tensor_free(single[i]);
    otensor *arrs[] = { m->ln1_w, m->ln1_b, m->qkv_w, m->qkv_b, m->proj_w, m->proj_b, m->ln2_w, m->ln2_b, m->fc1_w, m->fc1_b, m->fc2_w, m->fc2_b };
    for (int a = 0; a < 12; ++a) { if (arrs[a]) for (int i = 0; i < m->L; ++i) tensor_free(&arrs[a][i]); free(arrs[a]); }
    for (int i = 0; i < m->n_labels; ++i) free(m->labels[i]);
    free(m->labels); free(m->label_keys); free(m);
}

int oracle_model_in_chans(const omodel *m) { return m->Cin; }
int oracle_model_out_rows(const omodel *m) { return m->R; }
void oracle_model_hparams(const omodel *m, int *out7) {
    out7[0] = m->D; out7[1] = m->L; out7[2] = m->H; out7[3] = m->C; out7[4] = m->P; out7[5] = m->S; out7[6] = m->ftype;
}

/* ------------------------------------------------------------------------ ops */
/* ggml_compute_forward_norm_f32 + ggml_mul + ggml_add (vit.cpp:808-812,881-885,915-919):
 * mean and variance accumulated in double, scale = 1/sqrtf(var+eps), then *w, +b as
 * separate f32 ops (no FMA: file is built with -ffp-contract=off). */
void oracle_layernorm(const float *x, const float *w, const float *b, float *y, int rows, int D, float eps) {
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; ++r) {
        const float *xr = x + (size_t)r * D; float *yr = y + (size_t)r * D;
        double sum = 0.0;
        for (int i = 0; i < D; ++i) sum += (double)xr[i];
        const float mean = (float)(sum / D);
        double sum2 = 0.0;
        for (int i = 0; i < D; ++i) { float v = xr[i] - mean; yr[i] = v; sum2 += (double)(v * v); }
        const float variance = (float)(sum2 / D);
        const float scale = 1.0f / sqrtf(variance + eps);
        for (int i = 0; i < D; ++i) { float t = yr[i] * scale; t = t * w[i]; yr[i] = t + b[i]; }
    }
}

/* ggml_mul_mat(W, x) + bias (vit.cpp:820-821,868-869,889-890,896-897,927-928):
 * y[m][n] = dot(W[n][:], x'[m][:]) + bias[n], x' = x converted to W's vec_dot type. */
void oracle_linear(const otensor *W, const float *bias, const float *x, float *y, int M, const omode *md) {
    const int K = (int)W->ne[0], Nn = (int)W->ne[1];
    const int quant = (W->type != T_F32 && W->type != T_F16) && md->quant_act;
    const int ar = (W->type == T_F32 && md->act_round == 1) ? 0 : md->act_round;  /* f32 weights: ggml keeps x in f32 */
    float *wbuf = NULL;
    if (md->w_round == 2 && !quant) {
        wbuf = (float *)malloc((size_t)K * Nn * 4);
        for (size_t i = 0; i < (size_t)K * Nn; ++i) wbuf[i] = round_bf16(W->f32[i]);
    }
    const float *Wf = wbuf ? wbuf : W->f32;
#pragma omp parallel
    {
        g_dot_exact = md->dot_exact;
        float *xr = (float *)malloc((size_t)K * 4);
        blk_q8 *xq = quant ? (blk_q8 *)malloc(sizeof(blk_q8) * (K / QK)) : NULL;
#pragma omp for schedule(static)
        for (int m = 0; m < M; ++m) {
            const float *xm = x + (size_t)m * K; float *ym = y + (size_t)m * Nn;
            if (quant) {
                if (md->quant_act == 2) { for (int k = 0; k < K; ++k) xr[k] = round_sel(xm[k], ar); xm = xr; }   /* probe: the blocks formed from the ROUNDED activation (what a device path that stores 16-bit activations can do) */
                if (W->type == T_Q4_1 || W->type == T_Q5_1) quant_q8_1(xm, xq, K); else quant_q8_0(xm, xq, K);
                const size_t rb = (size_t)(K / QK) * type_block_bytes(W->type);
                for (int n = 0; n < Nn; ++n) { float v = dot_quant(W->type, W->raw + rb * n, xq, K); ym[n] = bias ? v + bias[n] : v; }
            } else {
                for (int k = 0; k < K; ++k) xr[k] = round_sel(xm[k], ar);
                for (int n = 0; n < Nn; ++n) { float v = dot_f32(Wf + (size_t)n * K, xr, K); ym[n] = bias ? v + bias[n] : v; }
            }
        }
        free(xr); free(xq);
    }
    free(wbuf);
}

/* ggml_vec_gelu_f32: y = FP16_TO_FP32(table_gelu_f16[FP32_TO_FP16(x)]) (vit.cpp:893). */
void oracle_gelu(float *x, size_t n, int lut) {
    init_tables();
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) {
        if (lut == 1) x[i] = f16_to_f32(g_table_gelu[f32_to_f16(x[i])]);
        else if (lut == 2) x[i] = round_bf16(gelu_f32(x[i]));
        else x[i] = gelu_f32(x[i]);
    }
}

/* ggml_compute_forward_soft_max_f32 (vit.cpp:856,931): max, e_i via the fp16 exp LUT
 * on fp16(x_i - max), sum in double, scale by (float)(1/sum). */
void oracle_softmax_rows(float *x, int rows, int n, int lut) {
    init_tables();
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; ++r) {
        float *p = x + (size_t)r * n;
        float mx = -INFINITY;
        for (int i = 0; i < n; ++i) if (p[i] > mx) mx = p[i];
        double sum = 0.0;
        for (int i = 0; i < n; ++i) {
            float val;
            if (p[i] == -INFINITY) val = 0.0f;
            else if (lut == 1) val = f16_to_f32(g_table_exp[f32_to_f16(p[i] - mx)]);
            else if (lut == 2) val = round_bf16(expf(round_bf16(p[i] - mx)));
            else val = expf(p[i] - mx);
            sum += (double)val; p[i] = val;
        }
        const float inv = (float)(1.0 / sum);
        for (int i = 0; i < n; ++i) p[i] *= inv;
    }
}

/* Attention for one image (vit.cpp:826-866): S[q][key] = dot_f32(K[key],Q[q]) * 0.125,
 * softmax over keys, O[q][d] = dot_f32(V^T[d][:], P[q][:]).  qkv row = [q(D)|k(D)|v(D)],
 * each D split head-major [h][64] (vit.cpp:826,834). */
static void attention_image(const float *qkv, float *out, int N, int D, int H, const omode *md) {
    const int d = D / H; const float scale = 1.0f / sqrtf((float)d);
    const int qkr = md->attn_qk_round >= 0 ? md->attn_qk_round : md->attn_round;
    const int vr = md->attn_v_round >= 0 ? md->attn_v_round : md->attn_round;
    g_dot_exact = md->dot_exact;
    float *Q = (float *)malloc((size_t)N * d * 4), *Kb = (float *)malloc((size_t)N * d * 4);
    float *VT = (float *)malloc((size_t)N * d * 4), *S = (float *)malloc((size_t)N * N * 4);
    for (int h = 0; h < H; ++h) {
        for (int t = 0; t < N; ++t) for (int e = 0; e < d; ++e) {
            const float *row = qkv + (size_t)t * 3 * D;
            Q[t * d + e] = round_sel(row[h * d + e], qkr);
            Kb[t * d + e] = round_sel(row[D + h * d + e], qkr);
            VT[e * N + t] = round_sel(row[2 * D + h * d + e], vr);
        }
        for (int q = 0; q < N; ++q) for (int k = 0; k < N; ++k) S[(size_t)q * N + k] = dot_f32(Kb + k * d, Q + q * d, d) * scale;
        /* softmax over keys; with attn_round the GPU feeds UNNORMALISED e_i (already fp16 under lut=1) to the PV product */
        if (md->attn_round == 0) {
            for (int q = 0; q < N; ++q) {
                float *p = S + (size_t)q * N; float mx = -INFINITY;
                for (int i = 0; i < N; ++i) if (p[i] > mx) mx = p[i];
                double sum = 0.0;
                for (int i = 0; i < N; ++i) {
                    float val = md->lut == 1 ? f16_to_f32(g_table_exp[f32_to_f16(p[i] - mx)])
                              : md->lut == 2 ? round_bf16(expf(round_bf16(p[i] - mx))) : expf(p[i] - mx);
                    sum += (double)val; p[i] = val;
                }
                const float inv = (float)(1.0 / sum);
                for (int i = 0; i < N; ++i) p[i] *= inv;
                for (int e = 0; e < d; ++e) out[(size_t)q * D + h * d + e] = dot_f32(VT + (size_t)e * N, p, N);
            }
        } else {
            for (int q = 0; q < N; ++q) {
                float *p = S + (size_t)q * N; float mx = -INFINITY;
                for (int i = 0; i < N; ++i) if (p[i] > mx) mx = p[i];
                float sum = 0.0f;
                for (int i = 0; i < N; ++i) {
                    float val = md->lut == 1 ? f16_to_f32(g_table_exp[f32_to_f16(p[i] - mx)])
                              : md->lut == 2 ? round_bf16(expf(p[i] - mx)) : round_sel(expf(p[i] - mx), md->attn_round);
                    sum += val; p[i] = val;
                }
                const float inv = 1.0f / sum;
                for (int e = 0; e < d; ++e) out[(size_t)q * D + h * d + e] = dot_f32(VT + (size_t)e * N, p, N) * inv;
            }
        }
    }
    free(Q); free(Kb); free(VT); free(S);
}

void oracle_attention(const float *qkv, float *out, int n_img, int N, int D, int H, const omode *md) {
    init_tables();
#pragma omp parallel for schedule(dynamic)
    for (int b = 0; b < n_img; ++b) attention_image(qkv + (size_t)b * N * 3 * D, out + (size_t)b * N * D, N, D, H, md);
}

/* Patch embedding (vit.cpp:747-797): HWC f32 -> planar (759-768) -> ggml_conv_2d_sk_p0
 * = im2col to fp16 with k = c*P*P + ky*P + kx, dot with the fp16 kernel in f32, + bias
 * (773-775), token t = px + g*py (778-791), cls row prepended (794), + pos_embed (797). */
void oracle_patch_embed(const omodel *m, const float *img_hwc, float *X, int n_img, const omode *md) {
    const int D = m->D, P = m->P, S = m->S, g = m->g, N = m->N, Cin = m->Cin, K = Cin * P * P;   /* ViTSTR: one grey plane (vitstr.cpp:713-731) */
    /* ggml's im2col always emits fp16 (kernel must be f16, vit.cpp:515) -> act rounding is at least fp16 unless 'ideal' */
    const int ar = md->act_round;
    float *wbuf = NULL;
    if (md->w_round == 2) { wbuf = (float *)malloc((size_t)K * D * 4); for (size_t i = 0; i < (size_t)K * D; ++i) wbuf[i] = round_bf16(m->pe_w.f32[i]); }
    const float *W = wbuf ? wbuf : m->pe_w.f32;
#pragma omp parallel
    {
        g_dot_exact = md->dot_exact;
        float *col = (float *)malloc((size_t)K * 4);
#pragma omp for schedule(static) collapse(2)
        for (int b = 0; b < n_img; ++b) for (int t = 0; t < g * g; ++t) {
            const int py = t / g, px = t % g;
            const float *img = img_hwc + (size_t)b * S * S * Cin;
            for (int c = 0; c < Cin; ++c) for (int ky = 0; ky < P; ++ky) for (int kx = 0; kx < P; ++kx)
                col[c * P * P + ky * P + kx] = round_sel(img[((size_t)(py * P + ky) * S + (px * P + kx)) * Cin + c], ar);
            float *xr = X + ((size_t)b * N + 1 + t) * D;
            for (int n = 0; n < D; ++n) { float v = dot_f32(W + (size_t)n * K, col, K); v = v + m->pe_b.f32[n]; xr[n] = v + m->pos.f32[(size_t)(1 + t) * D + n]; }
        }
        free(col);
    }
    for (int b = 0; b < n_img; ++b) for (int n = 0; n < D; ++n) X[(size_t)b * N * D + n] = m->cls.f32[n] + m->pos.f32[n];
    free(wbuf);
}

/* One encoder layer in place on X [rows=n_img*N][D] (vit.cpp:802-901). */
void oracle_layer(const omodel *m, int il, float *X, int n_img, const omode *md) {
    const int D = m->D, N = m->N, M = n_img * N;
    float *U = (float *)malloc((size_t)M * D * 4);
    float *QKV = (float *)malloc((size_t)M * 3 * D * 4);
    float *O = (float *)malloc((size_t)M * D * 4);
    float *Hh = (float *)malloc((size_t)M * 4 * D * 4);
    oracle_layernorm(X, m->ln1_w[il].f32, m->ln1_b[il].f32, U, M, D, m->eps);
    oracle_linear(&m->qkv_w[il], m->qkv_b[il].f32, U, QKV, M, md);
    oracle_attention(QKV, O, n_img, N, D, m->H, md);
    oracle_linear(&m->proj_w[il], m->proj_b[il].f32, O, U, M, md);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < (size_t)M * D; ++i) X[i] = U[i] + X[i];          /* vit.cpp:873 */
    oracle_layernorm(X, m->ln2_w[il].f32, m->ln2_b[il].f32, U, M, D, m->eps);
    oracle_linear(&m->fc1_w[il], m->fc1_b[il].f32, U, Hh, M, md);
    oracle_gelu(Hh, (size_t)M * 4 * D, md->lut);
    oracle_linear(&m->fc2_w[il], m->fc2_b[il].f32, Hh, U, M, md);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < (size_t)M * D; ++i) X[i] = U[i] + X[i];          /* vit.cpp:900 */
    free(U); free(QKV); free(O); free(Hh);
}

/* cls pooling + final LN + head + softmax (vit.cpp:910-933).  A ViTSTR file keeps the first R = 25 tokens of every image instead of
 * the cls token alone (extensions/vitstr.cpp/vitstr.cpp:864-904): logits / probs are [n_img * R][C], row = image * R + token. */
void oracle_head(const omodel *m, const float *X, float *logits, float *probs, int n_img, const omode *md) {
    const int D = m->D, N = m->N, C = m->C, R = m->R, rows = n_img * R;
    float *cls = (float *)malloc((size_t)rows * D * 4), *z = (float *)malloc((size_t)rows * D * 4);
    for (int b = 0; b < n_img; ++b) memcpy(cls + (size_t)b * R * D, X + (size_t)b * N * D, (size_t)R * D * 4);
    oracle_layernorm(cls, m->norm_w.f32, m->norm_b.f32, z, rows, D, m->eps);
    oracle_linear(&m->head_w, m->head_b.f32, z, logits, rows, md);
    if (probs) { memcpy(probs, logits, (size_t)rows * C * 4); oracle_softmax_rows(probs, rows, C, md->lut); }
    free(cls); free(z);
}

/* Full forward (vit_encode_image, vit.cpp:718-941) for n_img images given as the
 * normalised HWC f32 arrays vit_image_preprocess emits.  x_dump (optional) receives
 * the residual stream after patch-embed and after every layer: [(L+1)][n_img*N][D]. */
int oracle_forward(const omodel *m, const float *img_hwc, int n_img, const omode *md,
                   float *logits, float *probs, float *x_dump) {
    const int D = m->D, N = m->N; const size_t XN = (size_t)n_img * N * D;
    float *X = (float *)malloc(XN * 4);
    if (!X) return 1;
    init_tables();
    oracle_patch_embed(m, img_hwc, X, n_img, md);
    if (x_dump) memcpy(x_dump, X, XN * 4);
    for (int il = 0; il < m->L; ++il) {
        oracle_layer(m, il, X, n_img, md);
        if (x_dump) memcpy(x_dump + (size_t)(il + 1) * XN, X, XN * 4);
    }
    oracle_head(m, X, logits, probs, n_img, md);
    free(X);
    return 0;
}

/* ------------------------------------------------------------------ preprocess */
/* vit_image_preprocess_bicubic (vit.cpp:204-287): direct resize with scale nx/S, no
 * half-pixel offset, 4x4 neighbourhood with edge clamp, cubic coefficients evaluated
 * in double (the -1.0/3 etc. literals) then stored to float, polynomial in float;
 * only the jj==3 iteration's Cc survives (270-280); round, clamp, u8, (v-mean)/std. */
static inline int clipi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }
void oracle_preprocess_bicubic(const uint8_t *src, int nx, int ny, int S, float *dst) {
    const float m3[3] = {123.675f, 116.280f, 103.530f}, s3[3] = {58.395f, 57.120f, 57.375f};
    const float tx = (float)nx / (float)S, ty = (float)ny / (float)S;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < S; ++i) for (int j = 0; j < S; ++j) {
        const int x = (int)(tx * j), y = (int)(ty * i);
        const float dx = tx * j - x, dy = ty * i - y;
        for (int k = 0; k < 3; ++k) {
            float C[4];
            for (int jj = 0; jj <= 3; ++jj) {
                const int yy = clipi(y - 1 + jj, 0, ny - 1);
                const float p0 = src[((size_t)yy * nx + clipi(x - 1, 0, nx - 1)) * 3 + k];
                const float p1 = src[((size_t)yy * nx + clipi(x, 0, nx - 1)) * 3 + k];
                const float p2 = src[((size_t)yy * nx + clipi(x + 1, 0, nx - 1)) * 3 + k];
                const float p3 = src[((size_t)yy * nx + clipi(x + 2, 0, nx - 1)) * 3 + k];
                const float d0 = p0 - p1, d2 = p2 - p1, d3 = p3 - p1, a0 = p1;
                const float a1 = (float)(-1.0 / 3 * d0 + d2 - 1.0 / 6 * d3);
                const float a2 = (float)(1.0 / 2 * d0 + 1.0 / 2 * d2);
                const float a3 = (float)(-1.0 / 6 * d0 - 1.0 / 2 * d2 + 1.0 / 6 * d3);
                C[jj] = a0 + a1 * dx + a2 * dx * dx + a3 * dx * dx * dx;
            }
            const float d0 = C[0] - C[1], d2 = C[2] - C[1], d3 = C[3] - C[1], a0 = C[1];
            const float a1 = (float)(-1.0 / 3 * d0 + d2 - 1.0 / 6 * d3);
            const float a2 = (float)(1.0 / 2 * d0 + 1.0 / 2 * d2);
            const float a3 = (float)(-1.0 / 6 * d0 - 1.0 / 2 * d2 + 1.0 / 6 * d3);
            const float Cc = a0 + a1 * dy + a2 * dy * dy + a3 * dy * dy * dy;
            const uint8_t v = (uint8_t)fminf(fmaxf(roundf(Cc), 0.0f), 255.0f);
            dst[((size_t)i * S + j) * 3 + k] = ((float)v - m3[k]) / s3[k];
        }
    }
}

/* vit_image_preprocess_bilinear (vit.cpp:130-196): half-pixel centres. */
void oracle_preprocess_bilinear(const uint8_t *src, int nx, int ny, int S, float *dst) {
    const float m3[3] = {123.675f, 116.280f, 103.530f}, s3[3] = {58.395f, 57.120f, 57.375f};
    const float xs = nx / (float)S, ys = ny / (float)S;
    const int nx3 = (int)(nx / xs + 0.5f), ny3 = (int)(ny / ys + 0.5f);
    for (int y = 0; y < ny3; ++y) for (int x = 0; x < nx3; ++x) for (int c = 0; c < 3; ++c) {
        const float sx = (x + 0.5f) * xs - 0.5f, sy = (y + 0.5f) * ys - 0.5f;
        int x0 = (int)floorf(sx); if (x0 < 0) x0 = 0;
        int y0 = (int)floorf(sy); if (y0 < 0) y0 = 0;
        const int x1 = x0 + 1 < nx - 1 ? x0 + 1 : nx - 1, y1 = y0 + 1 < ny - 1 ? y0 + 1 : ny - 1;
        const float dx = sx - x0, dy = sy - y0;
        const float v00 = src[3 * ((size_t)y0 * nx + x0) + c], v01 = src[3 * ((size_t)y0 * nx + x1) + c];
        const float v10 = src[3 * ((size_t)y1 * nx + x0) + c], v11 = src[3 * ((size_t)y1 * nx + x1) + c];
        const float v0 = v00 * (1.0f - dx) + v01 * dx, v1 = v10 * (1.0f - dx) + v11 * dx;
        const float v = v0 * (1.0f - dy) + v1 * dy;
        const uint8_t v2 = (uint8_t)fminf(fmaxf(roundf(v), 0.0f), 255.0f);
        dst[3 * ((size_t)y * nx3 + x) + c] = ((float)v2 - m3[c]) / s3[c];
    }
}

/* vit_image_preprocess of the ViTSTR extension (extensions/vitstr.cpp/vitstr.cpp:128-201): grey = (uint8)(0.299 r + 0.587 g + 0.114 b)
 * in double, direct resize with scale n / S, 2x2 linear blend anchored at the truncated source coordinate (clamped to n - 2),
 * (v / 255 - 0.5) * 2; the padding loops start at res.nx == S and do nothing.  dst: [S][S] f32, one channel. */
void oracle_preprocess_vitstr(const uint8_t *src, int nx, int ny, int S, float *dst) {
    uint8_t *grey = (uint8_t *)malloc((size_t)nx * ny);
    for (size_t i = 0; i < (size_t)nx * ny; ++i) grey[i] = (uint8_t)(0.299 * src[3 * i] + 0.587 * src[3 * i + 1] + 0.114 * src[3 * i + 2]);
    const float x_scale = (float)nx / S, y_scale = (float)ny / S;
    for (int y = 0; y < S; ++y) for (int x = 0; x < S; ++x) {
        const float gx = x * x_scale, gy = y * y_scale;
        const int gxi = (int)gx, gyi = (int)gy;
        const float u = gx - gxi, v = gy - gyi;
        int px0 = gxi < nx - 2 ? gxi : nx - 2; if (px0 < 0) px0 = 0;
        int py0 = gyi < ny - 2 ? gyi : ny - 2; if (py0 < 0) py0 = 0;
        float val = (1 - u) * (1 - v) * grey[(size_t)py0 * nx + px0] + u * (1 - v) * grey[(size_t)py0 * nx + px0 + 1] +
                    (1 - u) * v * grey[(size_t)(py0 + 1) * nx + px0] + u * v * grey[(size_t)(py0 + 1) * nx + px0 + 1];
        dst[(size_t)y * S + x] = (val / 255.0f - 0.5f) * 2.0f;
    }
    free(grey);
}

/* ------------------------------------------------------------- small utilities */
/* Expose the dense f32 expansion of a named tensor (tests feed it to GPU kernels). */
const float *oracle_tensor_f32(omodel *m, const char *name, int64_t *ne4) {
    otensor *t = model_slot(m, name);
    if (!t || !t->f32) return NULL;
    if (ne4) memcpy(ne4, t->ne, 4 * sizeof(int64_t));
    return t->f32;
}
const otensor *oracle_tensor(omodel *m, const char *name) { return model_slot(m, name); }
void oracle_linear_named(omodel *m, const char *wname, const char *bname, const float *x, float *y, int M, const omode *md) {
    const otensor *W = model_slot(m, wname); const otensor *b = bname ? model_slot(m, bname) : NULL;
    oracle_linear(W, b ? b->f32 : NULL, x, y, M, md);
}
float oracle_round_f16(float x) { return round_f16(x); }
float oracle_round_bf16(float x) { return round_bf16(x); }
int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
/* Threads of the following calls (the reference's own knob: vit_params.n_threads, default 4 -- /root/reference/vit.h:95-103);
 * returns the previous value. */
int oracle_set_num_threads(int n) {
#ifdef _OPENMP
    const int prev = omp_get_max_threads();
    if (n > 0) omp_set_num_threads(n);
    return prev;
#else
    (void)n; return 1;
#endif
}
/* quantize_row_q4_0_reference etc. are restated in vit.cpp_amd's own quantize tool
 * tests; the oracle only needs the decode side. */

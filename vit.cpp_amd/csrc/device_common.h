// device_common.h -- types and helpers shared by the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace vitx {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define GPTR(p) ((const __attribute__((address_space(1))) void *)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void *)(p))

template <typename T> struct Elem;
template <> struct Elem<_Float16> {
    typedef half8 v8; typedef half4 v4;
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ f32x4 mfma16(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
    typedef short bits4 __attribute__((ext_vector_type(4)));      // half the k-slots of mfma16: lane group g holds k = 4 g .. 4 g + 3
    static __device__ __forceinline__ f32x4 mfma16k16(bits4 a, bits4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(v4, a), __builtin_bit_cast(v4, b), c, 0, 0, 0); }
};
template <> struct Elem<__bf16> {
    typedef bf16x8 v8; typedef bf16x4 v4;
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ f32x4 mfma16(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
    typedef short bits4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ f32x4 mfma16k16(bits4 a, bits4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0); }
};
template <typename T> __device__ __forceinline__ float rnd(float x) { return (float)(T)x; }   // round-trip through the operand type

// Packed pair helpers for the softmax inner loop (gfx950: v_cvt_pk_{f16,bf16}_f32, v_fma_mix_f32, v_dot2c_f32_{f16,bf16}).
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <typename T> struct Pair;
template <> struct Pair<_Float16> {
    typedef _Float16 v2 __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ float sum2(v2 p, float acc) { return __builtin_amdgcn_fdot2(p, v2{(_Float16)1.0f, (_Float16)1.0f}, acc, false); }
};
template <> struct Pair<__bf16> {
    typedef __bf16 v2 __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ float sum2(v2 p, float acc) { return __builtin_amdgcn_fdot2_f32_bf16(p, v2{(__bf16)1.0f, (__bf16)1.0f}, acc, false); }
};
// (T)a, (T)b rounded to nearest even in one instruction
template <typename T> __device__ __forceinline__ typename Pair<T>::v2 round_pair(float a, float b) { return __builtin_convertvector((f32x2{a, b}), typename Pair<T>::v2); }

// tanh-GELU of the reference (ggml_gelu_f32): 0.5*x*(1+tanh(sqrt(2/pi)*x*(1+0.044715*x*x))),
// evaluated as x*sigmoid(2u) = x / (1 + exp(-2u)), algebraically identical and stable in both tails.
// With u = c*x*(1 + a*x^2): exp(-2u) = exp2(x * (B + A*x^2)), B = -2*c*log2(e), A = B*a -- 3 multiplies, 1 fma, 1 add,
// v_exp_f32 and v_rcp_f32 per element (the straightforward form needs 7 multiplies; the GELU epilogue is VALU-bound).
// Saturates correctly: x -> -inf gives exp2(+inf) = inf, rcp = 0, result -0; x -> +inf gives exp2(-inf) = 0, result x.
__device__ __forceinline__ float gelu_tanh(float x) {
    constexpr float B = -2.0f * 0.79788456080286535588f * 1.44269504088896340736f;
    constexpr float A = B * 0.044715f;
    const float p = __builtin_fmaf(x * x, A, B);
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * p));
}

// two elements at a time: the multiplies / fma / add become v_pk_*_f32 (exp and rcp stay scalar, quarter rate)
__device__ __forceinline__ f32x2 gelu_tanh2(f32x2 x) {
    constexpr float B = -2.0f * 0.79788456080286535588f * 1.44269504088896340736f;
    constexpr float A = B * 0.044715f;
    const f32x2 p = __builtin_elementwise_fma(x * x, f32x2{A, A}, f32x2{B, B});
    const f32x2 t = x * p;
    const f32x2 d = f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} + f32x2{1.0f, 1.0f};
    return x * f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
}

// fc1 epilogue activation of two adjacent outputs (ggml_gelu through the fp16 table, /root/reference/vit.cpp:893).
//   F16 (parity mode): the table's semantics -- argument rounded to fp16, result rounded to fp16.
//   BF16 (the dtype BASELINE names): there is no bf16 rounding point in the reference to reproduce, so the argument stays f32 and only
//   the stored value is rounded (it is the next GEMM's operand): one convert and two unpacks per pair less, same tolerance band.
template <typename T> __device__ __forceinline__ typename Pair<T>::v2 gelu_out_pair(float v0, float v1);
template <> __device__ __forceinline__ Pair<_Float16>::v2 gelu_out_pair<_Float16>(float v0, float v1) {
    const Pair<_Float16>::v2 p = round_pair<_Float16>(v0, v1);
    const f32x2 y = gelu_tanh2(f32x2{(float)p[0], (float)p[1]});
    return round_pair<_Float16>(y[0], y[1]);
}
template <> __device__ __forceinline__ Pair<__bf16>::v2 gelu_out_pair<__bf16>(float v0, float v1) {
    const f32x2 y = gelu_tanh2(f32x2{v0, v1});
    return round_pair<__bf16>(y[0], y[1]);
}

// Softmax numerators of the attention kernels for two adjacent keys (ggml_soft_max, /root/reference/vit.cpp:856).
//   F16 (parity mode): e = round(exp(round(s/8 - max/8))) -- the fp16 exp table's semantics; nmx = -max * kScale, kScale = 1/8 (exact).
//   BF16: e = round_bf16(exp2(s * log2(e)/8 - max * log2(e)/8)): one fma and one v_exp_f32 per key, no rounded exponent.
template <typename T> struct AttnExp;
template <> struct AttnExp<_Float16> {
    static constexpr float kScale = 0.125f;
    static __device__ __forceinline__ Pair<_Float16>::v2 pair(float s0, float s1, float nmx) {
        const f32x2 d = __builtin_elementwise_fma(f32x2{s0, s1}, f32x2{0.125f, 0.125f}, f32x2{nmx, nmx});       // one v_pk_fma_f32: same bits as two fmaf
        const Pair<_Float16>::v2 dh = round_pair<_Float16>(d[0], d[1]);
        const f32x2 t = f32x2{(float)dh[0], (float)dh[1]} * f32x2{1.44269504f, 1.44269504f};
        return round_pair<_Float16>(__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1]));
    }
};
template <> struct AttnExp<__bf16> {
    static constexpr float kScale = 0.125f * 1.44269504088896340736f;
    static __device__ __forceinline__ Pair<__bf16>::v2 pair(float s0, float s1, float nmx) {
        const f32x2 d = __builtin_elementwise_fma(f32x2{s0, s1}, f32x2{kScale, kScale}, f32x2{nmx, nmx});        // one v_pk_fma_f32: same bits as two fmaf
        return round_pair<__bf16>(__builtin_amdgcn_exp2f(d[0]), __builtin_amdgcn_exp2f(d[1]));
    }
};

// The same with a run-time score scale 1 / sqrt(head_dim) (attention_generic_kernel: head dims other than 64; ggml_scale_inplace,
// /root/reference/vit.cpp:853).  k(scale) is what multiplies the raw score AND the row maximum (nmx = -max * k).
template <typename T> struct AttnExpRt;
template <> struct AttnExpRt<_Float16> {
    static __device__ __forceinline__ float k(float scale) { return scale; }
    static __device__ __forceinline__ Pair<_Float16>::v2 pair(float s0, float s1, float nmx, float kk) {
        const f32x2 d = __builtin_elementwise_fma(f32x2{s0, s1}, f32x2{kk, kk}, f32x2{nmx, nmx});
        const Pair<_Float16>::v2 dh = round_pair<_Float16>(d[0], d[1]);
        const f32x2 t = f32x2{(float)dh[0], (float)dh[1]} * f32x2{1.44269504f, 1.44269504f};
        return round_pair<_Float16>(__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1]));
    }
};
template <> struct AttnExpRt<__bf16> {
    static __device__ __forceinline__ float k(float scale) { return scale * 1.44269504088896340736f; }
    static __device__ __forceinline__ Pair<__bf16>::v2 pair(float s0, float s1, float nmx, float kk) {
        const f32x2 d = __builtin_elementwise_fma(f32x2{s0, s1}, f32x2{kk, kk}, f32x2{nmx, nmx});
        return round_pair<__bf16>(__builtin_amdgcn_exp2f(d[0]), __builtin_amdgcn_exp2f(d[1]));
    }
};

// Reductions over the four 16-lane rows of a wave (lanes l, l ^ 16, l ^ 32, l ^ 48) without the LDS crossbar: v_permlane16_swap /
// v_permlane32_swap (gfx950) exchange rows in the VALU -- no ds_bpermute, no lgkmcnt wait in the middle of a dependent chain.
// swap16(x, x) leaves {row 0, row 0, row 2, row 2} and {row 1, row 1, row 3, row 3}; swap32 of the result {lower half, lower half} and
// {upper half, upper half}: after both steps every lane holds the combination of all four rows.
// (elements are copied out before the bit cast: __builtin_bit_cast applied to a[1] of the returned vector reads element 0 -- hipcc 7.2)
__device__ __forceinline__ void rows_swap16(float x, float &u, float &v) {
    const auto a = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, x), false, false);
    const unsigned a0 = a[0], a1 = a[1];
    u = __builtin_bit_cast(float, a0); v = __builtin_bit_cast(float, a1);
}
__device__ __forceinline__ void rows_swap32(float x, float &u, float &v) {
    const auto a = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, x), false, false);
    const unsigned a0 = a[0], a1 = a[1];
    u = __builtin_bit_cast(float, a0); v = __builtin_bit_cast(float, a1);
}
__device__ __forceinline__ float rows4_max(float x) {
    float u, v;
    rows_swap16(x, u, v); x = fmaxf(u, v);
    rows_swap32(x, u, v); return fmaxf(u, v);
}
__device__ __forceinline__ float rows4_sum(float x) {       // (row 0 + row 1) + (row 2 + row 3) in every lane: the same bits everywhere
    float u, v;
    rows_swap16(x, u, v); x = u + v;
    rows_swap32(x, u, v); return u + v;
}

// ------------------------------------------------------------------------------------------------
// LayerNorm statistics by 256-column tiles (ggml_norm, /root/reference/vit.cpp:808-812, 881-885): the ONE definition both the
// stand-alone kernel (kernels.hip) and the LayerNorm fused into the residual GEMMs (gemm_pp.hip) follow, operation for operation,
// so that a row's result does not depend on which of them produced it (batch-size independence of the whole forward).
//   tile c (columns 256 c ..): the row's 256 values are 64 pieces of 4 consecutive columns; piece id = 16 w + 8 j + k
//     a(piece)  = (x0 + x1) + (x2 + x3)
//     s(w, k)   = a(w, 0, k) + a(w, 1, k)
//     P(w)      = ln_sum8 over k = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7))      -- a butterfly: every k holds the same bits
//     S_c       = ((P(0) + P(1)) + P(2)) + P(3);   mean_c = S_c / 256
//     M2_c      = the same tree over (x - mean_c)^2  (two passes inside the tile: no cancellation)
//   row: ln_combine() merges the tiles in index order (Chan et al.: equal counts): mean = (sum of mean_c) / NT,
//     M2 = sum of M2_c + 256 * sum of (mean_c - mean)^2, rstd = 1 / sqrt(M2 / D + eps);  y = ((x - mean) * rstd) * w + b.
// ggml's own order (double accumulation over the whole row) differs from this by f32 rounding only: within one operand ulp of
// oracle.layernorm (tests/test_gpu_kernels.py, test_gpu_parity_r02.py).
// ------------------------------------------------------------------------------------------------
template <int CTRL> __device__ __forceinline__ float dpp_f32(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true)); }
// sum over the 8 lanes that share (lane >> 3); every one of them ends up with the same bits
__device__ __forceinline__ float ln_sum8(float v) {
    v = v + dpp_f32<0xB1>(v);       // quad_perm [1, 0, 3, 2]: k ^ 1
    v = v + dpp_f32<0x4E>(v);       // quad_perm [2, 3, 0, 1]: k ^ 2
    v = v + dpp_f32<0x141>(v);      // row_half_mirror: the other quad of the 8
    return v;
}
__device__ __forceinline__ float ln_piece_sum(f32x4 x) { return (x[0] + x[1]) + (x[2] + x[3]); }
__device__ __forceinline__ float ln_piece_sq(f32x4 x, float m) {
    const float d0 = x[0] - m, d1 = x[1] - m, d2 = x[2] - m, d3 = x[3] - m;
    return (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
}
constexpr int LN_MAX_TILES = 4;         // hidden sizes 256 .. 1024 take the tiled definition
__device__ __forceinline__ void ln_combine(const float (&mc)[LN_MAX_TILES], const float (&m2)[LN_MAX_TILES], int NT, int D, float eps, float &mean, float &rstd) {
    float sm = mc[0], q = m2[0];
#pragma unroll
    for (int c = 1; c < LN_MAX_TILES; ++c) if (c < NT) { sm = sm + mc[c]; q = q + m2[c]; }
    mean = sm / (float)NT;
    float dv = mc[0] - mean, w = dv * dv;
#pragma unroll
    for (int c = 1; c < LN_MAX_TILES; ++c) if (c < NT) { dv = mc[c] - mean; w = w + dv * dv; }
    rstd = 1.0f / sqrtf((q + 256.0f * w) / (float)D + eps);
}

// One row of NT * 256 values by one wave, the tiled definition above: lane l holds piece l of each tile (w = l >> 4, j = (l >> 3) & 1,
// k = l & 7): one fully coalesced 1 KiB load per tile.  Used by layernorm_kernel, layernorm_fixup_kernel (kernels.hip) and by the
// prologue of a GEMM that consumes rows a LayerNorm-fusing GEMM left to the fix-up (gemm_pp.hip).
template <typename T, int NT>
__device__ __forceinline__ void ln_row_tiled(const float *__restrict__ xr, const float *__restrict__ w, const float *__restrict__ b, T *__restrict__ yr, float eps, int lane) {
    f32x4 v[NT];
    float mc[LN_MAX_TILES] = {0.0f, 0.0f, 0.0f, 0.0f}, m2[LN_MAX_TILES] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int c = 0; c < NT; ++c) v[c] = *(const f32x4 *)(xr + c * 256 + lane * 4);
    auto tile_total = [&](float a) {       // a = this lane's piece value -> S_c (uniform)
        const float s = a + __shfl_xor(a, 8);                      // s(w, k) = a(w, 0, k) + a(w, 1, k)
        const float p = ln_sum8(s);                                // P(w), the same bits in the 16 lanes of wave column w
        const float p0 = __shfl(p, 0), p1 = __shfl(p, 16), p2 = __shfl(p, 32), p3 = __shfl(p, 48);
        return ((p0 + p1) + p2) + p3;
    };
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        mc[c] = tile_total(ln_piece_sum(v[c])) * (1.0f / 256.0f);
        m2[c] = tile_total(ln_piece_sq(v[c], mc[c]));
    }
    float mean, rstd;
    ln_combine(mc, m2, NT, NT * 256, eps, mean, rstd);
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        const int idx = c * 256 + lane * 4;
        const f32x4 ww = *(const f32x4 *)(w + idx), bb = *(const f32x4 *)(b + idx);
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { float t = (v[c][e] - mean) * rstd; t = t * ww[e]; o[e] = t + bb[e]; }
        const typename Pair<T>::v2 lo = round_pair<T>(o[0], o[1]), hi = round_pair<T>(o[2], o[3]);
        *(typename Elem<T>::v4 *)(yr + idx) = typename Elem<T>::v4{lo[0], lo[1], hi[0], hi[1]};
    }
}

// ------------------------------------------------------------------------------------------------
// LDS tile image shared by the GEMM and attention kernels: rows of 64 elements (128 B = 8 slots of
// 16 B).  Two rows form one 256-B bank line; the 16 slots of a line are XOR-ed with (line & 15) so a
// ds_read_b128 lane group (16 rows, same logical slot) touches 16 distinct slots: conflict-free.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int swz_byte(int row, int slot /*0..7*/) {
    const int line = row >> 1;
    const int s16 = ((row & 1) << 3) | slot;
    return line * 256 + ((s16 ^ (line & 15)) << 4);
}
// inverse: physical 16-B slot index p (within the tile) -> logical (row, slot)
__device__ __forceinline__ void swz_inv(int p, int &row, int &slot) {
    const int line = p >> 4;
    const int s16 = (p & 15) ^ (line & 15);
    row = line * 2 + (s16 >> 3);
    slot = s16 & 7;
}


}  // namespace vitx

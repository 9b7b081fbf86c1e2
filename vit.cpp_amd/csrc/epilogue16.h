// epilogue16.h -- fused epilogue shared by every GEMM kernel, for accumulators of v_mfma_f32_16x16x32_{f16,bf16}.
//
// All GEMM families compute C = A . W^T with SWAPPED products, acc = mfma16(W fragment, A fragment, acc): the 16 x 16 result tile
// then has, in lane (l15 = lane & 15, g4 = lane >> 4), register e = C[row l15][column 4 g4 + e] -- four CONSECUTIVE columns of one
// row, so a lane stores 8 bytes (f16 / bf16 outputs) or 16 bytes (f32 outputs) at a time and the residual read-modify-write is one
// dwordx4 load + one dwordx4 store.  Every family consumes K in the same order with the same instruction, so a given output
// element is the same bits whichever kernel the batch size selects (tests/test_gpu_parity_r02.py).
//
// Epilogues (reference: /root/reference/vit.cpp): EPI_BIAS qkv (:820-823); EPI_BIAS_GELU fc1 + ggml_gelu through the fp16
// table (:888-893; the value is rounded to the operand type before and after the activation); EPI_BIAS_RESID proj / fc2 +
// residual add in f32 (:868-873, :899-902); EPI_BIAS_F32 classifier head (:917-922); EPI_PATCH patch embedding + position
// embedding, patch row -> token row (:774-800).
#pragma once
#include "device_common.h"
#include "kernels.h"

namespace vitx {

// acc[t][u]: tile rows row0 + 16 t (row0 already includes l15), columns col0 + 16 u .. + 3 (col0 already includes 4 g4).
// FULL: the whole workgroup tile is inside [0, M_real) x [0, N) -- no bounds checks, vector bias loads.  Otherwise every element is
// checked on its own (N need not be a multiple of 4: the classifier head has as many columns as the model has classes).
template <typename T, int EPI, int TM, int TN, bool FULL>
__device__ __forceinline__ void epilogue16(const GemmArgs &g, f32x4 (&acc)[TM][TN], int row0, int col0) {
    typedef typename Elem<T>::v4 v4;
    typedef typename Pair<T>::v2 v2;
#pragma unroll
    for (int u = 0; u < TN; ++u) {
        const int c = col0 + u * 16;
        f32x4 bv = {0.0f, 0.0f, 0.0f, 0.0f};
        if (g.bias) {
            if (FULL) bv = *(const f32x4 *)(g.bias + c);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[e] = (c + e < g.N) ? g.bias[c + e] : 0.0f;
            }
        }
        const bool vec = FULL || (c + 3 < g.N && (g.ldo & 3) == 0);       // this lane's four columns exist and are 8 / 16-byte aligned
        if constexpr (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) {
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                const int row = row0 + t * 16;
                if (!FULL && row >= g.M_real) continue;
                const f32x4 v = acc[t][u] + bv;
                v2 p0 = round_pair<T>(v[0], v[1]), p1 = round_pair<T>(v[2], v[3]);
                if constexpr (EPI == EPI_BIAS_GELU) {      // round to the operand type (ggml's fp16 LUT input), tanh-GELU, round (LUT output)
                    const f32x2 y0 = gelu_tanh2(f32x2{(float)p0[0], (float)p0[1]}), y1 = gelu_tanh2(f32x2{(float)p1[0], (float)p1[1]});
                    p0 = round_pair<T>(y0[0], y0[1]); p1 = round_pair<T>(y1[0], y1[1]);
                }
                T *o = (T *)g.out + (size_t)row * g.ldo + c;
                if (vec) *(v4 *)o = v4{p0[0], p0[1], p1[0], p1[1]};
                else {
                    const T s[4] = {p0[0], p0[1], p1[0], p1[1]};
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (c + e < g.N) o[e] = s[e];
                }
            }
        } else {
            // f32 outputs.  The residual / position-embedding rows of all TM tiles are loaded BEFORE the first store: loads and
            // stores go through pointers the compiler must assume alias, so interleaving them would serialise TM round trips.
            float *o[TM]; f32x4 add[TM];
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                const int row = row0 + t * 16;
                add[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                const bool row_ok = FULL || row < g.M_real;
                if constexpr (EPI == EPI_PATCH) {           // patch row -> token row (+1 per image for the cls slot), + position embedding
                    const int b = row / g.tpi, tk = row - b * g.tpi;
                    o[t] = (float *)g.out + ((size_t)row + b + 1) * g.ldo + c;
                    const float *pe = g.pos + (size_t)(tk + 1) * g.ldo + c;
                    if (row_ok) {
                        if (vec) add[t] = *(const f32x4 *)pe;
                        else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) add[t][e] = (c + e < g.N) ? pe[e] : 0.0f;
                        }
                    }
                } else {
                    o[t] = (float *)g.out + (size_t)row * g.ldo + c;
                    if constexpr (EPI == EPI_BIAS_RESID) {
                        if (row_ok) {
                            if (vec) add[t] = *(const f32x4 *)o[t];
                            else {
#pragma unroll
                                for (int e = 0; e < 4; ++e) add[t][e] = (c + e < g.N) ? o[t][e] : 0.0f;
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                if (!FULL && row0 + t * 16 >= g.M_real) continue;
                f32x4 r = acc[t][u] + bv;
                if constexpr (EPI == EPI_BIAS_RESID || EPI == EPI_PATCH) r = r + add[t];      // (acc + bias) + x: the reference's order (vit.cpp:868-873)
                if (vec) *(f32x4 *)o[t] = r;
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (c + e < g.N) o[t][e] = r[e];
                }
            }
        }
    }
}

}  // namespace vitx

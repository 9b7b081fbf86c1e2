// kernels.h -- launchers of the hand-written gfx950 kernels (kernels.hip).
// All pointers are device pointers; every launcher only enqueues on `stream`.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vitx {

enum { DT_F16 = 0, DT_BF16 = 1 };

// GEMM epilogues (C = A[M][K] . W[N][K]^T, f32 accumulate)
enum {
    EPI_BIAS = 0,        // out(dtype)[m][n] = acc + bias[n]                       qkv      (vit.cpp:820-821)
    EPI_BIAS_GELU = 1,   // out(dtype)[m][n] = gelu_tanh(round(acc + bias[n]))     fc1      (vit.cpp:889-893)
    EPI_BIAS_RESID = 2,  // out(f32)[m][n]   = (acc + bias[n]) + out[m][n]         proj/fc2 (vit.cpp:868-873, 896-900)
    EPI_BIAS_F32 = 3,    // out(f32)[m][n]   = acc + bias[n]                       head     (vit.cpp:927-928)
    EPI_PATCH = 4        // out(f32)[m + m/tpi + 1][n] = (acc + bias[n]) + pos[(m%tpi + 1)][n]   (vit.cpp:772-797)
};

struct GemmArgs {
    const void *A; const void *W; const float *bias; void *out; const float *pos;
    int M;        // rows computed (multiple of the M tile; buffers are padded to it)
    int M_real;   // rows stored
    int N;        // columns stored (real)
    int N_pad;    // columns of W available (multiple of the N tile; zero rows beyond N)
    int K;        // multiple of 64
    int lda, ldw, ldo;
    int tpi;      // EPI_PATCH: patch tokens per image (g*g)
    int dbg;      // ablation bits for kernel experiments (VITX_GEMM_DBG): 1 no DMA in loop, 2 no ds_read in loop, 4 no MFMA, 8 no epilogue
};

hipError_t launch_gemm(int dtype, int epi, const GemmArgs &a, hipStream_t stream);
int gemm_tile_m();   // M granularity the GEMM needs (buffer row padding)
int gemm_tile_n();

// im2col of the f32 HWC image into dtype rows [n_img*g*g][Kpad], k = c*P*P + ky*P + kx (vit.cpp:759-772)
hipError_t launch_patchify(int dtype, const float *img, void *out, int n_img, int S, int P, int Kpad, int rows_pad, hipStream_t stream);
// X[b*N + 0][:] = cls + pos[0]  (vit.cpp:794-797)
hipError_t launch_cls_rows(const float *cls, const float *pos, float *X, int n_img, int N, int D, hipStream_t stream);
// y[r][:] (dtype) = LN(x[r*ldx ...]) * w + b   (vit.cpp:808-812)
hipError_t launch_layernorm(int dtype, const float *x, long ldx, const float *w, const float *b, void *y, long ldy, int M, int D, float eps, hipStream_t stream);
// fused per-(image,head) attention  (vit.cpp:826-866)
hipError_t launch_attention(int dtype, const void *qkv, void *out, int n_img, int N, int D, int H, hipStream_t stream);
// class softmax with the reference's fp16 (or bf16) exp rounding (vit.cpp:931)
hipError_t launch_softmax(int dtype, const float *logits, float *probs, int rows, int cols, int ld, hipStream_t stream);
hipError_t launch_preprocess(const void *u8, float *out, int n, int nx, int ny, int S, int bicubic, hipStream_t stream);

}  // namespace vitx

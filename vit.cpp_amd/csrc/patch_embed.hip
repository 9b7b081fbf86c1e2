// patch_embed.hip -- patch embedding as ONE kernel: the im2col gather, the GEMM, + bias + position embedding, the token scatter and the
// class-token rows (/root/reference/vit.cpp:747-797: HWC -> planar repack, ggml_conv_2d_sk_p0 = im2col to fp16 + mul_mat, + bias, concat
// with cls_token, + pos_embed).
//
// r01/r02 ran this as three launches -- patchify_kernel wrote the [patches][K] operand rows (77 MB at 256 images) for a GEMM to read back,
// cls_rows_kernel filled token 0.  Here the A tiles are gathered straight from the f32 HWC image on their way into the LDS:
//   * K is walked in the image's own memory order, k' = (ky * P + kx) * Cin + c (the weight matrix is permuted to that order at upload:
//     the sum over k is the same set of products, grouped differently -- f32 summation noise against ggml's channel-major order), so
//     16 consecutive k' are 64 contiguous bytes of one image row whenever P * Cin is a multiple of 16 (every patch-16 / patch-32 model, RGB
//     or the one-channel ViTSTR input); other geometries (patch 8 or 14) take a per-element gather;
//   * 128 x 128 x 64 tiles, 4 waves, the LDS image / fragment layout / products / epilogue of the v1 GEMM (kernels.hip): W by LDS-DMA, A by
//     registers -- each thread loads 2 x 16 floats of the next K-tile under the current K-tile's MFMAs, rounds them to the operand type
//     exactly where ggml's im2col emits fp16, and writes two 16-byte slots of the swizzled image;
//   * the epilogue (epilogue16.h, EPI_PATCH) adds bias and pos_embed[1 + patch], scatters to token row image * N + 1 + patch, and the
//     lanes that hold patch 0 of an image also write its class row cls_token + pos_embed[0].
// 0.7 % of the forward's FLOPs: a plain double-buffered kernel, not the persistent one.
#include "device_common.h"
#include "kernels.h"
#include "epilogue16.h"

namespace vitx {

namespace pe {
constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE = BM * BK * 2, STAGE = 2 * TILE, LDS_BYTES = 2 * STAGE;      // [A | W] x 2 buffers = 64 KiB
}

template <typename T>
__global__ __launch_bounds__(256) void patch_embed_kernel(GemmArgs g, const float *__restrict__ img, const float *__restrict__ cls, int S, int P, int Cin, int gsz /* patches per image row */) {
    using namespace pe;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename Elem<T>::v8 v8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g4 = lane >> 4;

    // XCD-aware tile order (v1 GEMM): blocks b, b + 8, .. of one XCD take consecutive tiles, which share the A row panel (n fastest)
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int ntn = g.N_pad / BN;
    const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;

    const T *W = (const T *)g.W;
    int woff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { int row, slot; swz_inv(i * 256 + tid, row, slot); woff[i] = (n0 + row) * g.ldw + slot * 8; }
    auto stage_w = [&](int buf, int k0) {
        char *base = smem + buf * STAGE + TILE + wave * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i) __builtin_amdgcn_global_load_lds(GPTR(W + woff[i] + k0), LPTR(base + i * 4096), 16, 0, 0);
    };
    // A gather: this thread owns tile row a_row and the 16-float groups 2 a_h, 2 a_h + 1 of every K-tile
    const int a_row = tid >> 1, a_h = tid & 1;
    const int K = Cin * P * P, PC = P * Cin, tpi = g.tpi;
    const int m = m0 + a_row;
    const bool row_ok = m < g.M_real;
    const float *prow;                  // first float of the patch: pixel (py * P, px * P), channel 0
    {
        const int b = m / tpi, t = m - b * tpi, py = t / gsz, px = t - py * gsz;
        prow = img + (((size_t)b * S + (size_t)py * P) * S + (size_t)px * P) * Cin;
    }
    const bool fast = (PC & 15) == 0;   // a 16-float group never straddles an image row
    f32x4 areg[2][4];
    auto load_a = [&](int kt) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int k0 = kt * BK + (a_h * 2 + q) * 16;
            if (fast) {
                const int ky = k0 / PC, r0 = k0 - ky * PC;
                const f32x4 *src = (const f32x4 *)(prow + (size_t)ky * S * Cin + r0);
                const bool ok = row_ok && k0 < K;            // K is a multiple of 16 here: a group is all inside or all padding
#pragma unroll
                for (int e = 0; e < 4; ++e) areg[q][e] = ok ? src[e] : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int k = k0 + e, ky = k / PC, r = k - ky * PC;
                    areg[q][e >> 2][e & 3] = (row_ok && k < K) ? prow[(size_t)ky * S * Cin + r] : 0.0f;
                }
            }
        }
    };
    auto write_a = [&](int buf) {       // rounded to the operand type here: ggml's im2col writes fp16 (vit.cpp:772)
        char *at = smem + buf * STAGE;
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const f32x4 lo = areg[q][2 * hh], hi = areg[q][2 * hh + 1];
                const typename Pair<T>::v2 p0 = round_pair<T>(lo[0], lo[1]), p1 = round_pair<T>(lo[2], lo[3]), p2 = round_pair<T>(hi[0], hi[1]), p3 = round_pair<T>(hi[2], hi[3]);
                *(v8 *)(at + swz_byte(a_row, (a_h * 2 + q) * 2 + hh)) = v8{p0[0], p0[1], p1[0], p1[1], p2[0], p2[1], p3[0], p3[1]};
            }
    };

    const int wm = wave >> 1, wn = wave & 1;
    f32x4 acc[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[t][u] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    int a_rd[4][2], w_rd[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            a_rd[t][k2] = swz_byte(wm * 64 + t * 16 + l15, k2 * 4 + g4);
            w_rd[t][k2] = TILE + swz_byte(wn * 64 + t * 16 + l15, k2 * 4 + g4);
        }

    const int nk = g.K / BK;
    stage_w(0, 0);
    load_a(0); write_a(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) { stage_w(cur ^ 1, (kt + 1) * BK); load_a(kt + 1); }
        const char *sb = smem + cur * STAGE;
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            v8 af[4], wf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) { af[t] = *(const v8 *)(sb + a_rd[t][k2]); wf[t] = *(const v8 *)(sb + w_rd[t][k2]); }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[t][u] = Elem<T>::mfma16(wf[u], af[t], acc[t][u]);
        }
        if (kt + 1 < nk) write_a(cur ^ 1);       // the other buffer: every wave finished reading it before the previous barrier
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    const bool full = (m0 + BM <= g.M_real) && (n0 + BN <= g.N);
    const int row0 = m0 + wm * 64 + l15, col0 = n0 + wn * 64 + 4 * g4;
    if (full) epilogue16<T, EPI_PATCH, 4, 4, true>(g, acc, row0, col0);
    else epilogue16<T, EPI_PATCH, 4, 4, false>(g, acc, row0, col0);
    // class rows (vit.cpp:794-797): token 0 of image b = cls_token + pos_embed[0]; written by the lanes that hold patch 0 of that image
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int row = row0 + t * 16;
        if (row >= g.M_real || row % tpi) continue;
        float *o = (float *)g.out + ((size_t)row + row / tpi) * g.ldo;          // token row b * (tpi + 1)
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const int c = col0 + u * 16 + e; if (c < g.N) o[c] = cls[c] + g.pos[c]; }
    }
}

// Weight matrix for the kernel above: w_perm[n][(ky * P + kx) * Cin + c] = w[n][c * P * P + ky * P + kx] (host side, at upload)
void patch_embed_permute_k(const uint16_t *w, uint16_t *w_perm, int N, int Cin, int P, int k_pad) {
    const int K = Cin * P * P;
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < Cin; ++c)
            for (int y = 0; y < P; ++y)
                for (int x = 0; x < P; ++x) w_perm[(size_t)n * k_pad + (y * P + x) * Cin + c] = w[(size_t)n * k_pad + c * P * P + y * P + x];
    (void)K;
}

hipError_t launch_patch_embed(int dtype, const float *img, const void *w_perm, const float *bias, const float *pos, const float *cls, float *X,
                              int n_img, int S, int P, int Cin, int D, int n_pad, int k_pad, hipStream_t stream, bool prepare) {
    if (prepare) {
        hipError_t e = hipFuncSetAttribute((const void *)patch_embed_kernel<_Float16>, hipFuncAttributeMaxDynamicSharedMemorySize, pe::LDS_BYTES);
        if (e != hipSuccess) return e;
        return hipFuncSetAttribute((const void *)patch_embed_kernel<__bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, pe::LDS_BYTES);
    }
    const int gsz = S / P, tpi = gsz * gsz;
    if (n_img <= 0 || P <= 0 || S % P || n_pad % pe::BN || k_pad % pe::BK || k_pad < Cin * P * P || D % 4) return hipErrorInvalidValue;
    GemmArgs g{};
    g.W = w_perm; g.bias = bias; g.out = X; g.pos = pos;
    g.M_real = n_img * tpi; g.M = (g.M_real + pe::BM - 1) / pe::BM * pe::BM; g.N = D; g.N_pad = n_pad; g.K = k_pad; g.ldw = k_pad; g.ldo = D; g.tpi = tpi;
    const int grid = (g.M / pe::BM) * (n_pad / pe::BN);
    if (dtype == DT_F16) hipLaunchKernelGGL(patch_embed_kernel<_Float16>, dim3(grid), dim3(256), pe::LDS_BYTES, stream, g, img, cls, S, P, Cin, gsz);
    else hipLaunchKernelGGL(patch_embed_kernel<__bf16>, dim3(grid), dim3(256), pe::LDS_BYTES, stream, g, img, cls, S, P, Cin, gsz);
    return hipGetLastError();
}

}  // namespace vitx

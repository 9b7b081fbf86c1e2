// kernels.hip -- hand-written CDNA4 (gfx950) kernels of the ViT forward path.
//
// Written for MI355X only: 64-lane wavefronts, v_mfma_f32_16x16x32_{f16,bf16} (GEMMs) / v_mfma_f32_32x32x16 (attention),
// global_load_lds (LDS-DMA) staging with a source-side XOR swizzle, 160 KiB LDS.
// The math each kernel implements is the ggml op sequence vit_encode_image emits
// (/root/reference/vit.cpp:718-941); per-kernel citations below.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <memory>
#include <mutex>
#include <type_traits>
#include <vector>

#include "kernels.h"
#include "epilogue16.h"
#include "device_common.h"

namespace vitx {

// ------------------------------------------------------------------------------------------------
// GEMM  C[M][N] = A[M][K] . W[N][K]^T  (ggml_mul_mat, vit.cpp:820,868,889,896,927 and the im2col GEMM
// of ggml_conv_2d_sk_p0, vit.cpp:772) with the bias / GELU / residual / pos-embed epilogues fused.
// 128x128x64 tile, 4 waves (2x2), each wave 64x64 = 4x4 tiles of MFMA 16x16x32, LDS double buffer filled
// by global_load_lds dwordx4 (one K-tile ahead).
// ------------------------------------------------------------------------------------------------
constexpr int GBM = 128, GBN = 128, GBK = 64;
constexpr int G_TILE_BYTES = GBM * GBK * 2;            // 16 KiB per operand tile
constexpr int G_STAGE_BYTES = 2 * G_TILE_BYTES;        // A + W
constexpr int G_LDS_BYTES = 2 * G_STAGE_BYTES;         // double buffer: 64 KiB

// Q4 = true: W stays in ggml q4_0 block form in HBM (GemmArgs::W = nibble plane, ::Wscale = f16 block scales, 4.5 bits per weight) and
// is expanded in the LDS-fill path: every thread loads ONE block (16 B of nibbles + its scale) of the next K-tile into registers
// while the current K-tile is multiplied, then writes (q - 8) * d, rounded once to the operand type exactly as the host-side
// expansion does (HostTensor::decode_f32 + RNE), into the same swizzled LDS image the LDS-DMA path produces.  The reference keeps
// quantised weights through compute the same way (ggml_mul_mat on a q4_0 src0: /root/reference/vit.cpp:645-678, 820).
typedef unsigned q4_u32x4 __attribute__((ext_vector_type(4)));
template <typename T, int EPI, bool Q4 = false>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g4 = lane >> 4;

    // XCD-aware tile order: blocks b, b+8, b+16, ... (same XCD, co-resident) get consecutive tile ids,
    // which share the same A row panel (n fastest).  Bijective for any grid size.
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int ntn = g.N_pad / GBN;
    const int m0 = (tile / ntn) * GBM, n0 = (tile % ntn) * GBN;

    const T *A = (const T *)g.A, *W = (const T *)g.W;
    // per-thread source offsets of the 4+4 16-byte pieces this thread DMA-loads per K tile
    int aoff[4], woff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int row, slot; swz_inv(i * 256 + tid, row, slot);
        aoff[i] = (m0 + row) * g.lda + slot * 8;
        woff[i] = (n0 + row) * g.ldw + slot * 8;
    }
    auto stage = [&](int buf, int k0) {
        char *base = smem + buf * G_STAGE_BYTES + wave * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds(GPTR(A + aoff[i] + k0), LPTR(base + i * 4096), 16, 0, 0);
            if constexpr (!Q4) __builtin_amdgcn_global_load_lds(GPTR(W + woff[i] + k0), LPTR(base + G_TILE_BYTES + i * 4096), 16, 0, 0);
        }
    };
    // q4_0 path: thread -> (tile row tid / 2, block tid % 2 of the 64-deep K-tile)
    const int q_row = tid >> 1, q_kb = tid & 1;
    const int q_nbk = g.K >> 5;
    const unsigned char *q_qs = (const unsigned char *)g.W + (size_t)(n0 + q_row) * q_nbk * 16;
    const uint16_t *q_d = g.Wscale + (size_t)(n0 + q_row) * q_nbk;
    q4_u32x4 q_regs = {0, 0, 0, 0}; uint16_t q_scale = 0;
    auto load_q4 = [&](int kt) {
        const int b = kt * 2 + q_kb;
        q_regs = *(const q4_u32x4 *)(q_qs + (size_t)b * 16);
        q_scale = q_d[b];
    };
    auto write_q4 = [&](int buf) {
        const float d = (float)__builtin_bit_cast(_Float16, q_scale);
        char *wt = smem + buf * G_STAGE_BYTES + G_TILE_BYTES;
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {        // block elements 8 sl .. 8 sl + 7: low nibbles of bytes 0-15 first, then the high nibbles (block_q4_0)
            typename Elem<T>::v8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int byte = (sl & 1) * 8 + e;
                const unsigned w = q_regs[byte >> 2] >> ((byte & 3) * 8);
                const int nib = (sl < 2) ? (int)(w & 15u) : (int)((w >> 4) & 15u);
                v[e] = (T)((float)(nib - 8) * d);
            }
            *(typename Elem<T>::v8 *)(wt + swz_byte(q_row, q_kb * 4 + sl)) = v;
        }
    };

    const int wm = wave >> 1, wn = wave & 1;
    f32x4 acc[4][4];                                  // the wave's 64 x 64 block as 4 x 4 tiles of v_mfma_f32_16x16x32 (epilogue16.h)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[t][u] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    // fragment read addresses: row = w*64 + t*16 + l15, 16-byte slot = k2*4 + g4 (k-step k2 = 32 of the K-tile's 64)
    int a_rd[4][2], w_rd[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            a_rd[t][k2] = swz_byte(wm * 64 + t * 16 + l15, k2 * 4 + g4);
            w_rd[t][k2] = G_TILE_BYTES + swz_byte(wn * 64 + t * 16 + l15, k2 * 4 + g4);
        }

    const int nk = g.K / GBK;
    stage(0, 0);
    if constexpr (Q4) { load_q4(0); write_q4(0); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) { stage(cur ^ 1, (kt + 1) * GBK); if constexpr (Q4) load_q4(kt + 1); }
        const char *sb = smem + cur * G_STAGE_BYTES;
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            typename Elem<T>::v8 af[4], wf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                af[t] = *(const typename Elem<T>::v8 *)(sb + a_rd[t][k2]);
                wf[t] = *(const typename Elem<T>::v8 *)(sb + w_rd[t][k2]);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[t][u] = Elem<T>::mfma16(wf[u], af[t], acc[t][u]);
        }
        if constexpr (Q4) { if (kt + 1 < nk) write_q4(cur ^ 1); }     // the other buffer: every wave finished reading it before the previous barrier
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // every wave is past the last K-tile's barrier: the staging buffers are free, each wave takes 4 KiB as its epilogue patch
    const bool full = (m0 + GBM <= g.M_real) && (n0 + GBN <= g.N);
    epilogue16_tile<T, EPI, 2>(g, acc, full, m0, n0, wm * 64, wn * 64, smem + wave * 4096, lane);
}

int gemm_tile_m() { return 256; }   // row padding of every activation buffer (ring kernel tile height)
int gemm_tile_n() { return GBN; }

template <typename T, bool Q4 = false>
static hipError_t launch_gemm_t(int epi, const GemmArgs &a, hipStream_t stream, bool prepare) {
    const int grid = prepare ? 1 : (a.M / GBM) * (a.N_pad / GBN);
    const dim3 blk(256);
#define VITX_GEMM_CASE(E)                                                                                   \
    case E: {                                                                                               \
        if (prepare) return hipFuncSetAttribute((const void *)gemm_nt_kernel<T, E, Q4>, hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS_BYTES); \
        hipLaunchKernelGGL((gemm_nt_kernel<T, E, Q4>), dim3(grid), blk, G_LDS_BYTES, stream, a);            \
    } break;
    switch (epi) {
        VITX_GEMM_CASE(EPI_BIAS)
        VITX_GEMM_CASE(EPI_BIAS_GELU)
        VITX_GEMM_CASE(EPI_BIAS_RESID)
        VITX_GEMM_CASE(EPI_BIAS_F32)
        VITX_GEMM_CASE(EPI_PATCH)
        VITX_GEMM_CASE(EPI_BIAS_HILO)
    default: return hipErrorInvalidValue;
    }
#undef VITX_GEMM_CASE
    return hipGetLastError();
}

// Kernel selection (t.gemm_cfg: an explicit family, set by vitx_op_gemm_ex for the parity tests).
//   * >= 128 tiles of 256x256 and K % 128 == 0: the ping-pong persistent kernel (gemm_pp.hip);
//   * otherwise 128x256 ring tiles, or the skinny 64x128 ring kernel when those would leave half of the CUs idle
//     (a handful of images: same K order per element, so results stay bit-identical across batch sizes);
//   * a column count that is a multiple of 128 but not of 256: the skinny ring tiles at any row count.
static int wide_ring_cfg(const GemmArgs &a) { return gemm_ring_supports(a, 945) ? 945 : 445; }

// Persistent grid of the ping-pong kernel.  Tiles are dealt round-robin, so with one workgroup per CU a partial last round
// (e.g. 339 tiles = 256 + 83) leaves most of the chip idle while the first round ran at the power-capped clock.  Balanced:
// rounds = ceil(tiles / CUs), grid = ceil(tiles / rounds) rounded up to the 8 XCDs -- every workgroup walks the same number
// of tiles, fewer CUs are lit at a higher clock (the GEMM is energy-bound: half the CUs deliver 76 % of the throughput),
// and the CUs left free take the other sub-batch's kernels.
static int pp_grid(const Tuning &t, const GemmArgs &a) {
    int cap = t.n_cu & ~7;
    if (cap <= 0) cap = 256;
    const long ntiles = (long)(a.M / 256) * (a.N_pad / 256);
    if (!t.gemm_balance || ntiles <= cap) return cap;
    const long rounds = (ntiles + cap - 1) / cap;
    const long g = ((ntiles + rounds - 1) / rounds + 7) & ~7L;
    return (int)(g < cap ? g : cap);
}

static bool is_wide(const GemmArgs &a) {
    const long t256 = (long)(a.M / 256) * (a.N_pad / 256);
    return a.M % 256 == 0 && a.N_pad % 256 == 0 && t256 >= 128 && (gemm_pp_supports(a) || gemm_ring_supports(a, 445));
}
int gemm_pp_ln_grid(int n_cu, int M, int N);      // gemm_pp.hip
int gemm_ln_grid(int n_cu, int M, int N) { return gemm_pp_ln_grid(n_cu, M, N); }
bool gemm_ln_fusable(const Tuning &t, const GemmArgs &a) {
    // the peer mapping of pp_epilogue_ln (bid & 7 = XCD, whole row blocks per XCD) is built for the 8 XCDs of an MI355X in SPX mode: any
    // other partitioning (CPX / a different part) takes the stand-alone LayerNorm -- same bits, no spinning on peers that are elsewhere
    return t.n_xcd == 8 && t.n_cu % 8 == 0 && t.gemm_cfg < 0 && !t.gemm_split && !t.pp_flags && a.M > 0 && is_wide(a) && gemm_pp_supports(a) && a.N == a.ldo && a.N == a.N_pad && a.N % 256 == 0 &&
           a.N / 256 <= LN_MAX_TILES && (size_t)a.M * a.ldo * 4 < 0xf0000000u;
}
bool gemm_fix_capable(const Tuning &t, const GemmArgs &a) {
    return t.gemm_cfg < 0 && !t.gemm_split && !t.pp_flags && a.M > 0 && is_wide(a) && gemm_pp_supports(a) && a.K == a.lda && a.K % 256 == 0 && a.K / 256 <= LN_MAX_TILES;
}
static hipError_t launch_wide(const Tuning &t, int dtype, int epi, const GemmArgs &a0, hipStream_t stream) {
    GemmArgs a = a0; a.group_m = t.group_m;
    // Raster of the qkv / fc1 launches (r05, profiles/r05/raster_sweep.txt): an XCD walks groups of group_m row blocks x all column tiles.  While the
    // whole weight matrix fits beside the A panels in the XCD's 4 MiB L2 (+ a little: ViT-B qkv 3.4 MiB, fc1 4.5 MiB), group_m = 1 -- every CU of the
    // XCD on the same few row blocks, W resident, A streamed once -- is 0.4-0.7 % of the ViT-B forward faster than 8 (two independent A/Bs, same
    // bits); with ViT-L's matrices (6 / 8 MiB) it is 0.6 % slower, so they keep the A-resident groups of 8.
    if (!a.group_m && !a.ln && (epi == EPI_BIAS || epi == EPI_BIAS_HILO || epi == EPI_BIAS_GELU)) a.group_m = ((size_t)a.N_pad * a.K * 2 <= ((size_t)5 << 20)) ? 1 : 8;
    if (a.ln) return (epi == EPI_BIAS_RESID && gemm_ln_fusable(t, a)) ? launch_gemm_pp(dtype, epi, a, t.n_cu, stream, 0) : hipErrorInvalidValue;
    if (gemm_pp_supports(a)) return launch_gemm_pp(dtype, epi, a, pp_grid(t, a), stream, t.pp_flags);
    return launch_gemm_ring(t, dtype, epi, a, wide_ring_cfg(a), stream);
}

hipError_t launch_gemm(const Tuning &t, int dtype, int epi, const GemmArgs &a, hipStream_t stream) {
    if (a.M <= 0) return hipErrorInvalidValue;
    if (a.ln && !gemm_ln_fusable(t, a)) return hipErrorInvalidValue;      // the caller asks gemm_ln_fusable first
    if (a.fix && !gemm_fix_capable(t, a)) return hipErrorInvalidValue;     // ... and gemm_fix_capable
    int cfg = t.gemm_cfg;
    if (cfg == 1) return launch_gemm_pp(dtype, epi, a, pp_grid(t, a), stream, t.pp_flags);
    if (cfg > 1) return gemm_ring_supports(a, cfg) ? launch_gemm_ring(t, dtype, epi, a, cfg, stream) : hipErrorInvalidValue;
    if (is_wide(a)) {
        // Tail split (t.gemm_split; off: with the persistent kernel the second launch costs 5 % of the step, profiles/r02_forward_sweeps.txt,
        // re-measured with the 16x16x32 kernels in r02f): rows that fill whole rounds keep 256x256 tiles, the remaining rows are re-tiled
        // 128x256 (half-cost tiles) in a second launch.  Kept reachable through vitx_op_gemm_ex(kernel 2) so the path stays tested.
        const int ntm = a.M / 256, ntn = a.N_pad / 256;
        const long tiles = (long)ntm * ntn, rounds = tiles / t.n_cu, rem = tiles % t.n_cu;
        if (t.gemm_split && epi != EPI_PATCH && rounds >= 1 && rem > 0 && rem <= t.n_cu * 6 / 10) {
            const int m_main = (int)((rounds * t.n_cu) / ntn);                 // m-tiles that fit in whole rounds
            const int rows_main = m_main * 256;
            GemmArgs head = a, tail = a;
            head.M = rows_main; head.M_real = std::min(a.M_real, rows_main);
            const size_t esz_out = (epi == EPI_BIAS || epi == EPI_BIAS_GELU || epi == EPI_BIAS_HILO) ? 2 : 4;
            tail.A = (const char *)a.A + (size_t)rows_main * a.lda * 2;
            tail.out = (char *)a.out + (size_t)rows_main * a.ldo * esz_out;
            tail.M = a.M - rows_main; tail.M_real = a.M_real - rows_main;
            if (m_main >= 1 && rows_main < a.M && gemm_ring_supports(tail, 245)) {
                hipError_t e = launch_wide(t, dtype, epi, head, stream);
                if (e != hipSuccess) return e;
                if (tail.M_real <= 0) return hipSuccess;
                return launch_gemm_ring(t, dtype, epi, tail, 245, stream);
            }
        }
        return launch_wide(t, dtype, epi, a, stream);
    }
    cfg = 245;
    if ((long)(a.M / 128) * (a.N_pad / 256) < t.skinny_tiles && gemm_ring_supports(a, 122)) cfg = 122;
    if (gemm_ring_supports(a, cfg)) return launch_gemm_ring(t, dtype, epi, a, cfg, stream);
    // a column count that is a multiple of 128 but not of 256: the 64 x 128 ring tiles at any row count (the v1 128 x 128 kernel used to
    // take these; it now exists only in its q4_0 form, launch_gemm_q4)
    if (gemm_ring_supports(a, 122)) return launch_gemm_ring(t, dtype, epi, a, 122, stream);
    return hipErrorInvalidValue;
}

bool gemm_q4_supports(const GemmArgs &a) { return a.Wscale && a.M > 0 && a.M % GBM == 0 && a.N_pad % GBN == 0 && a.K % GBK == 0; }
hipError_t launch_gemm_q4(int dtype, int epi, const GemmArgs &a, hipStream_t stream) {
    if (!gemm_q4_supports(a)) return hipErrorInvalidValue;
    return dtype == DT_F16 ? launch_gemm_t<_Float16, true>(epi, a, stream, false) : launch_gemm_t<__bf16, true>(epi, a, stream, false);
}

// ------------------------------------------------------------------------------------------------
// LayerNorm (ggml_norm + ggml_mul + ggml_add_inplace, vit.cpp:808-812, 881-885, 915-919):
// mean, then biased variance of (x-mean), y = ((x-mean) * 1/sqrt(var+eps)) * w + b, rounded to the
// operand type of the GEMM that consumes it.  One wave per row, row kept in registers.
// Hidden sizes that are 1..4 tiles of 256 columns (256, 512, 768, 1024: every model the wide GEMMs run) take the TILED statistics
// of device_common.h, the definition the LayerNorm fused into the residual GEMMs (gemm_pp.hip) follows too: a row gets the same
// bits whichever of the two produced it.  Lane l of the wave holds piece l of each tile (w = l >> 4, j = (l >> 3) & 1, k = l & 7):
// one fully coalesced 1 KiB load per tile.
// ------------------------------------------------------------------------------------------------
template <typename T, int VEC, int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(const float *__restrict__ x, long ldx, const float *__restrict__ w, const float *__restrict__ b,
                                                        T *__restrict__ y, long ldy, int M, float eps, int group, long gstride) {
    constexpr int D = 64 * VEC * NV;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    // input row: group == 1 -> row * ldx; otherwise rows come in groups (row / group) * gstride + (row % group) * ldx
    // (the first `group` tokens of every image: the ViTSTR head, vitstr.cpp:864-883)
    const float *xr = group == 1 ? x + (size_t)row * ldx : x + (size_t)(row / group) * gstride + (size_t)(row % group) * ldx;
    T *yr = y + (size_t)row * ldy;
    if constexpr (VEC == 4 && NV <= LN_MAX_TILES) { ln_row_tiled<T, NV>(xr, w, b, yr, eps, lane); return; }
    float v[NV][VEC];
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = (i * 64 + lane) * VEC;
        if constexpr (VEC == 4) { const float4 t = *(const float4 *)(xr + idx); v[i][0] = t.x; v[i][1] = t.y; v[i][2] = t.z; v[i][3] = t.w; }
        else if constexpr (VEC == 2) { const float2 t = *(const float2 *)(xr + idx); v[i][0] = t.x; v[i][1] = t.y; }
        else v[i][0] = xr[idx];
#pragma unroll
        for (int j = 0; j < VEC; ++j) sum += v[i][j];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum / (float)D;
    float sum2 = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int j = 0; j < VEC; ++j) { v[i][j] -= mean; sum2 += v[i][j] * v[i][j]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum2 += __shfl_xor(sum2, o);
    const float scale = 1.0f / sqrtf(sum2 / (float)D + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = (i * 64 + lane) * VEC;
        T o[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) { float t = v[i][j] * scale; t = t * w[idx + j]; o[j] = (T)(t + b[idx + j]); }
        if constexpr (VEC == 4) *(typename Elem<T>::v4 *)(yr + idx) = typename Elem<T>::v4{o[0], o[1], o[2], o[3]};
        else {
#pragma unroll
            for (int j = 0; j < VEC; ++j) yr[idx + j] = o[j];
        }
    }
}

// Row blocks a LayerNorm-fusing GEMM left behind (GemmLn: todo[rb] == epoch): 64 workgroups, every wave takes one row of each such
// block -- no single-CU tail.  With nothing to do (the normal case) a workgroup reads the flags and exits.
template <typename T, int NT>
__global__ __launch_bounds__(256) void layernorm_fixup_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ b, T *__restrict__ y,
                                                              int n_blocks, float eps, const unsigned *__restrict__ todo, unsigned epoch) {
    const int lane = threadIdx.x & 63, wv = blockIdx.x * 4 + (threadIdx.x >> 6), nwv = gridDim.x * 4;
    // 64 flags per pass, one per lane (r03a: a serial scan of the ~200 flags made this launch 16 us with nothing to do, 0.5 ms per forward)
    for (int base = 0; base < n_blocks; base += 64) {
        const int rb_l = base + lane;
        const bool hit = rb_l < n_blocks && __builtin_nontemporal_load(todo + rb_l) == epoch;
        unsigned long long mask = __ballot(hit);
        while (mask) {
            const int rb = base + __builtin_ctzll(mask);
            mask &= mask - 1;
            for (int r = wv; r < 256; r += nwv) {
                const size_t row = (size_t)rb * 256 + r;
                ln_row_tiled<T, NT>(x + row * (NT * 256), w, b, y + row * (NT * 256), eps, lane);
            }
        }
    }
}
hipError_t launch_layernorm_fixup(int dtype, const float *x, const float *w, const float *b, void *y, int M, int D, float eps, const unsigned *todo, unsigned epoch, hipStream_t stream) {
    if (M % 256 || D % 256 || D / 256 < 1 || D / 256 > LN_MAX_TILES) return hipErrorInvalidValue;
    const dim3 grid(64), blk(256);
    const int nb = M / 256;
#define VITX_FIX_CASE(NT)                                                                                   \
    case NT:                                                                                                \
        if (dtype == DT_F16) hipLaunchKernelGGL((layernorm_fixup_kernel<_Float16, NT>), grid, blk, 0, stream, x, w, b, (_Float16 *)y, nb, eps, todo, epoch); \
        else hipLaunchKernelGGL((layernorm_fixup_kernel<__bf16, NT>), grid, blk, 0, stream, x, w, b, (__bf16 *)y, nb, eps, todo, epoch);                    \
        break;
    switch (D / 256) { VITX_FIX_CASE(1) VITX_FIX_CASE(2) VITX_FIX_CASE(3) VITX_FIX_CASE(4) default: return hipErrorInvalidValue; }
#undef VITX_FIX_CASE
    return hipGetLastError();
}

template <typename T>
static hipError_t launch_layernorm_t(const float *x, long ldx, const float *w, const float *b, void *y, long ldy, int M, int D, float eps, hipStream_t stream, int group, long gstride) {
    const dim3 grid((M + 3) / 4), blk(256);
#define VITX_LN_CASE(DD, VEC, NV) \
    case DD: hipLaunchKernelGGL((layernorm_kernel<T, VEC, NV>), grid, blk, 0, stream, x, ldx, w, b, (T *)y, ldy, M, eps, group, gstride); break;
    switch (D) {
        VITX_LN_CASE(64, 1, 1) VITX_LN_CASE(128, 2, 1) VITX_LN_CASE(192, 1, 3) VITX_LN_CASE(256, 4, 1) VITX_LN_CASE(384, 2, 3)
        VITX_LN_CASE(512, 4, 2) VITX_LN_CASE(768, 4, 3) VITX_LN_CASE(1024, 4, 4) VITX_LN_CASE(1280, 4, 5) VITX_LN_CASE(1536, 4, 6)
        // widths of other timm ViTs (SO400M 1152, ViT-g 1408, ViT-G 1664, ...) and of small test models
        VITX_LN_CASE(320, 1, 5) VITX_LN_CASE(448, 1, 7) VITX_LN_CASE(576, 1, 9) VITX_LN_CASE(640, 2, 5) VITX_LN_CASE(896, 2, 7)
        VITX_LN_CASE(1152, 2, 9) VITX_LN_CASE(1408, 2, 11) VITX_LN_CASE(1664, 2, 13) VITX_LN_CASE(2048, 4, 8)
    default: return hipErrorInvalidValue;
    }
#undef VITX_LN_CASE
    return hipGetLastError();
}
bool layernorm_supports(int D) {
    switch (D) { case 64: case 128: case 192: case 256: case 320: case 384: case 448: case 512: case 576: case 640: case 768: case 896: case 1024: case 1152: case 1280: case 1408: case 1536: case 1664: case 2048: return true; default: return false; }
}
hipError_t launch_layernorm(int dtype, const float *x, long ldx, const float *w, const float *b, void *y, long ldy, int M, int D, float eps, hipStream_t stream, int group, long gstride) {
    if (group < 1) return hipErrorInvalidValue;
    return dtype == DT_F16 ? launch_layernorm_t<_Float16>(x, ldx, w, b, y, ldy, M, D, eps, stream, group, gstride) : launch_layernorm_t<__bf16>(x, ldx, w, b, y, ldy, M, D, eps, stream, group, gstride);
}

// ------------------------------------------------------------------------------------------------
// Fused attention for one (image, head) per workgroup (vit.cpp:826-866): S = K Q^T * 1/8, softmax
// over keys, O = P V, heads merged on store.  head_dim is 64 for every model the reference converts.
//   * K [Nk][64] is staged in LDS in the swizzled row image above, V is staged TRANSPOSED
//     ([64][Nk+8], keys permuted inside each group of 16 so that the MFMA k-slot order of the P
//     registers needs no shuffle).
//   * "swapped" products: S^T = K . Q^T puts a whole score column (one query) in one lane pair, so the
//     softmax max/sum are in-register reductions plus one cross-half shuffle; O^T = V^T . P^T then
//     takes the probabilities straight from the accumulator registers as its B operand.
//   * each wave owns 32 queries; all NKT key tiles are kept in registers (single pass, no online
//     rescale), which fits N <= 608 tokens.
// exp follows ggml_soft_max: e = round(exp(round(s - max))) in the operand type (fp16 LUT in ggml).
// ------------------------------------------------------------------------------------------------
template <typename T, int NKT, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64, (NKT <= 9 && NWAVES <= 4) ? 2 : 1) void attention_kernel(const T *__restrict__ qkv, T *__restrict__ out, int N, int D, int H) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = NWAVES * 64;
    constexpr int NK = NKT * 32;          // padded key count
    constexpr int VLD = NK + 8;           // V^T row stride (elements); (VLD/8) odd -> conflict-free b128 reads
    char *Ks = smem;
    T *VT = (T *)(smem + NK * 128);
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const T *base = qkv + (size_t)b * N * 3 * D + h * 64;
    typedef typename Elem<T>::v8 v8;

    // ---- this wave's first query fragments: issued first so their latency hides under the K/V staging
    auto load_q = [&](int qt, v8 (&qf)[4]) {
        const int qrow = min(qt * 32 + l31, N - 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const v8 *)(base + (size_t)qrow * 3 * D + ks * 16 + hh * 8);
    };
    constexpr bool QPREF = NKT <= 9;      // longer sequences have no registers to spare for a prefetched Q tile
    v8 qf[4];
    if (QPREF && wave < NKT) load_q(wave, qf);

    // ---- stage K: 16-B pieces in row order (coalesced 128-B rows), all loads issued before the LDS writes
    {
        constexpr int IT = (NK * 8 + NT - 1) / NT;
        v8 kv[IT];
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int c = it * NT + tid, key = c >> 3, sl = c & 7;
#pragma unroll
            for (int j = 0; j < 8; ++j) kv[it][j] = (T)0.0f;
            if (c < NK * 8 && key < N) kv[it] = *(const v8 *)(base + (size_t)key * 3 * D + D + sl * 8);
        }
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int c = it * NT + tid;
            if (c < NK * 8) *(v8 *)(Ks + swz_byte(c >> 3, c & 7)) = kv[it];
        }
    }
    // ---- stage V^T: one work item = (key pair, 8 head dims).  Consecutive lanes take consecutive key pairs, so
    // each of the 8 transposed stores is a 4-byte (two keys) write to consecutive dwords of one V^T row (no bank
    // conflicts); keys 4-7 <-> 8-11 of every 16 are swapped (MFMA k-slot order of the P registers).
    {
        constexpr int NP = NK / 2, ITEMS = NP * 8, IT = (ITEMS + NT - 1) / NT;
        v8 va[IT], vb[IT];
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int c = it * NT + tid, pr = c % NP, sl = c / NP, key = 2 * pr;
#pragma unroll
            for (int j = 0; j < 8; ++j) { va[it][j] = (T)0.0f; vb[it][j] = (T)0.0f; }
            if (c < ITEMS && key < N) va[it] = *(const v8 *)(base + (size_t)key * 3 * D + 2 * D + sl * 8);
            if (c < ITEMS && key + 1 < N) vb[it] = *(const v8 *)(base + (size_t)(key + 1) * 3 * D + 2 * D + sl * 8);
        }
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int c = it * NT + tid, pr = c % NP, sl = c / NP, key = 2 * pr;
            if (c >= ITEMS) continue;
            const int a = key & 15, q4 = a >> 2, q4s = (q4 == 1) ? 2 : (q4 == 2) ? 1 : q4;
            const int pos = (key & ~15) | (q4s << 2) | (a & 3);
            typedef T v2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int j = 0; j < 8; ++j) *(v2 *)(VT + (sl * 8 + j) * VLD + pos) = v2{va[it][j], vb[it][j]};
        }
    }
    __syncthreads();

#pragma unroll 1
    for (int qt = wave; qt < NKT; qt += NWAVES) {
        int lds_off = 0;
        asm volatile("" : "+v"(lds_off));   // opaque zero: keeps the (query-independent) K / V^T fragment reads inside the loop instead of hoisted into ~220 live registers
        const int qrow = qt * 32 + l31;
        const bool qvalid = qrow < N;
        if (!QPREF) load_q(qt, qf);

        // S^T tiles: rows = keys, cols = queries
        f32x16 s[NKT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kt][r] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const v8 kf = *(const v8 *)(Ks + lds_off + swz_byte(kt * 32 + l31, ks * 2 + hh));
                s[kt] = Elem<T>::mfma(kf, qf[ks], s[kt]);
            }
        }
        if (QPREF && qt + NWAVES < NKT) load_q(qt + NWAVES, qf);       // next query tile of this wave: in flight during softmax + PV
        // mask padded keys, max of the raw scores; the 2^-3 scale is exact, so fma(s, 1/8, -max/8) == s/8 - max/8
        float mxs = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (kt == NKT - 1) {       // only the last key tile can hold padded keys
                    const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    if (key >= N) s[kt][r] = -INFINITY;
                }
                mxs = fmaxf(mxs, s[kt][r]);
            }
        mxs = fmaxf(mxs, __shfl_xor(mxs, 32));
        const float nmx = -AttnExp<T>::kScale * mxs;
        // e = round(exp(round(s/8 - max))) per ggml_soft_max, two keys per packed convert; exp(-inf) = 0 for padded keys.
        // The row sum adds the ROUNDED values (as ggml does) with one v_dot2c per pair.
        float sum = 0.0f;
        v8 p[NKT][2];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const typename Pair<T>::v2 eh = AttnExp<T>::pair(s[kt][r], s[kt][r + 1], nmx);
                sum = Pair<T>::sum2(eh, sum);
                p[kt][r >> 3][r & 7] = eh[0]; p[kt][r >> 3][(r & 7) + 1] = eh[1];
            }
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;

        // O^T = V^T . P^T : rows = head dims (2 tiles of 32), cols = queries
        f32x16 o[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] = 0.0f;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const v8 vf = *(const v8 *)((const char *)(VT + (dt * 32 + l31) * VLD + kt * 32 + half * 16 + hh * 8) + lds_off);
                    o[dt] = Elem<T>::mfma(vf, p[kt][half], o[dt]);
                }
        }
        if (qvalid) {
            T *orow = out + ((size_t)b * N + qrow) * D + h * 64;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    typename Elem<T>::v4 w4;
#pragma unroll
                    for (int j = 0; j < 4; ++j) w4[j] = (T)(o[dt][r4 * 4 + j] * inv);
                    *(typename Elem<T>::v4 *)(orow + dt * 32 + r4 * 8 + hh * 4) = w4;
                }
        }
    }
}

template <typename T, int NKT, int NWAVES>
static hipError_t launch_attention_inst(const void *qkv, void *out, int n_img, int N, int D, int H, hipStream_t stream) {
    constexpr int lds = NKT * 32 * 128 + 64 * (NKT * 32 + 8) * 2;
    if (n_img == 0) return hipFuncSetAttribute((const void *)attention_kernel<T, NKT, NWAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);   // device bring-up
    hipLaunchKernelGGL((attention_kernel<T, NKT, NWAVES>), dim3(n_img * H), dim3(NWAVES * 64), lds, stream, (const T *)qkv, (T *)out, N, D, H);
    return hipGetLastError();
}
template <typename T>
static hipError_t launch_attention_t(const void *qkv, void *out, int n_img, int N, int D, int H, hipStream_t stream) {
    const int nkt = (N + 31) / 32;
    switch (nkt) {
    case 1: return launch_attention_inst<T, 1, 1>(qkv, out, n_img, N, D, H, stream);
    case 2: return launch_attention_inst<T, 2, 2>(qkv, out, n_img, N, D, H, stream);
    case 3: return launch_attention_inst<T, 3, 3>(qkv, out, n_img, N, D, H, stream);
    case 4: return launch_attention_inst<T, 4, 4>(qkv, out, n_img, N, D, H, stream);
    case 5: return launch_attention_inst<T, 5, 4>(qkv, out, n_img, N, D, H, stream);
    case 6: return launch_attention_inst<T, 6, 4>(qkv, out, n_img, N, D, H, stream);
    // 197 tokens (224/16): 4 waves x 2 query tiles, two workgroups per CU (one stages K/V while the other computes): 109 us vs 124 us for
    // one 7-wave workgroup per CU on 256 x 12 heads (r01)
    case 7: return launch_attention_inst<T, 7, 4>(qkv, out, n_img, N, D, H, stream);
    case 9: return launch_attention_inst<T, 9, 4>(qkv, out, n_img, N, D, H, stream);      // 257 tokens (224/14)
    case 19: return launch_attention_inst<T, 19, 4>(qkv, out, n_img, N, D, H, stream);    // 577 tokens (384/16)
    default: return hipErrorInvalidValue;
    }
}
// ------------------------------------------------------------------------------------------------
// Pipelined two-pass attention (vit.cpp:826-866), any token count.  Same arithmetic as the two kernels above (same products,
// rounding points and summation order: bit-identical results), laid out for latency hiding instead of register residency:
//   * a workgroup = 4 waves = 4 query tiles of one (image, head); the query blocks of one (image, head) share an XCD (L2 hits
//     on the re-streamed K/V);
//   * keys stream through LDS in chunks of 64 (8 KiB K + 8 KiB V), double buffered, by LDS-DMA (buffer_load ... lds): no
//     staging registers and no VALU work -- the softmax's exp/convert instructions are what bounds this kernel.  K lands in the
//     swizzled row image (swizzle applied on the source side), V lands ROW-major and the V^T fragments of O^T = V^T P^T come
//     out of ds_read_b64_tr_b16 (a 16-lane group reads a [4 keys][16 dims] block and receives it transposed: lane i gets dim
//     i of keys 0..3, which is the k-slot order of the P registers -- tools/tr_probe.hip prints the mapping);
//   * 32 KiB of LDS and <= 128 VGPRs: four workgroups = 16 waves per CU, so the MFMA work of one wave runs under the VALU work
//     of the others (the single-pass kernel holds every score in registers: one wave per SIMD at 577 tokens).
// Pass 1 streams K for the row maxima, pass 2 streams K and V.  Keys past N inside the last chunk read the next image's rows
// (finite values; their scores are masked to -inf and their probabilities are exactly 0) or, past the end of the tensor, the
// zeros a buffer load returns out of range.
// ------------------------------------------------------------------------------------------------
// ONLINE (r04, bf16 only): ONE pass.  The reference's softmax goes through fp16 tables relative to the TRUE row maximum, which is why the F16
// builds take the maximum first; bf16 has no rounding point of the reference to reproduce there, and softmax is invariant under the per-row
// constant that is subtracted, so the bf16 build keeps a RUNNING maximum instead and never streams K a second time (64 images x 16 heads x 577
// tokens: 211 -> 160 us; ViT-L/16-384 forward +3.5 %, profiles/r04/ab_online_softmax.txt).  The constant is only moved when some row's tile maximum
// exceeds it by more than kTau (2^8 in the exponent: numerators stay <= 256, exact in bf16's range and harmless in the f32 sums) -- in practice
// during the first chunks only -- and then the accumulators and the running sum of every lane are rescaled by exp2 of its own shift.
// Rounding: P is rounded to bf16 at whatever scale it has (a relative rounding), O / sum once at the end, as before.
template <typename T, int FLAGS, bool ONLINE = false>      // FLAGS: ablation builds of tools/attn_bench.py (1 no re-staging, 2 no barriers, 4 no exp/convert, 8 no PV, 16 no pass 1); 0 = product
__global__ __launch_bounds__(256, 4) void attention_flow_kernel(const T *__restrict__ qkv, T *__restrict__ out, int N, int D, int H, int qblocks, int items, int n_img) {
    static_assert(!ONLINE || std::is_same<T, __bf16>::value, "the running-maximum schedule is the bf16 build's");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int CK = 64, KBYTES = CK * 128, VBYTES = CK * 128, BUF = KBYTES + VBYTES;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // blockIdx -> (item, query block): blocks whose index is equal mod 8 run on one XCD; an item's query blocks are consecutive there
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int item = (jb / qblocks) * 8 + xcd, qb = jb % qblocks;
    if (item >= items) return;
    const int b = item / H, h = item % H;
    const T *base = qkv + (size_t)b * N * 3 * D + h * 64;
    typedef typename Elem<T>::v8 v8;
    typedef short s4 __attribute__((ext_vector_type(4)));
    const int nch = (N + CK - 1) / CK;
    const int row_bytes = 3 * D * 2;

    const int qrow = (qb * 4 + wave) * 32 + l31;
    const bool qvalid = qrow < N;
    const bool wave_live = (qb * 4 + wave) * 32 < N;          // a wave past the last query tile only takes part in the barriers
    v8 qf[4];
    {
        const int qr = min(qrow, N - 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const v8 *)(base + (size_t)qr * 3 * D + ks * 16 + hh * 8);
    }

    // ---- LDS-DMA: physical 16-B piece p = it*256 + tid of a chunk image <-> (key row, 16-B slot) of the K / V column block
    const unsigned remaining = (unsigned)min((size_t)0xf0000000u, ((size_t)(n_img - b) * N * 3 * D - h * 64) * 2);
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)remaining, 0x00020000);
    int koff[2], voff[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int p = it * 256 + tid;
        int rr, sl; swz_inv(p, rr, sl);
        koff[it] = rr * row_bytes + D * 2 + sl * 16;
        const int vr = p >> 3, vs = (p & 7) ^ (((vr >> 1) & 1) << 2);          // V image: 16-B slot ^ 4 on rows 2, 3 (mod 4)
        voff[it] = vr * row_bytes + 2 * D * 2 + vs * 16;
    }
    auto stage = [&](char *buf, int key0, bool with_v) {
        char *dst = buf + wave * 1024;
        const int so = key0 * row_bytes;
#pragma unroll
        for (int it = 0; it < 2; ++it) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(dst + it * 4096), 16, koff[it], so, 0, 0);
        if (with_v) {
#pragma unroll
            for (int it = 0; it < 2; ++it) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(dst + KBYTES + it * 4096), 16, voff[it], so, 0, 0);
        }
    };
    // V^T fragment addresses: lane of a 16-lane group g = (lane >> 4) & 1 (dims 16g..16g+15 of the 32-dim tile) supplies the
    // address of key row (lane & 15) >> 2 (+ 4 hh), dims 4 (lane & 3)..+3; the tile's dt bit and the row's swizzle bit share bit 6
    int vrd[2];
    {
        const int r = 4 * hh + ((lane & 15) >> 2), rb = (r >> 1) & 1;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) vrd[dt] = KBYTES + r * 128 + ((dt ^ rb) << 6) + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;
    }
    // K fragment addresses inside a chunk image: tile kt adds kt * 4096 (the swizzle term depends on l31 only)
    int krd[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) krd[ks] = swz_byte(l31, ks * 2 + hh);
    auto qk_tile = [&](const char *cur, int kt, f32x16 &s) {             // S^T tile = K tile . Q^T (4 MFMAs)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) s = Elem<T>::mfma(*(const v8 *)(cur + krd[ks] + kt * 4096), qf[ks], s);
    };
    auto mask_tile = [&](int kt, int key0, f32x16 &s) {                   // keys >= N of the last chunk: -inf
        if (key0 + kt * 32 + 32 > N) {
#pragma unroll
            for (int r = 0; r < 16; ++r) if (key0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh >= N) s[r] = -INFINITY;
        }
    };
    auto max_tile = [&](const f32x16 &s, float &m) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) m = fmaxf(fmaxf(s[r], s[r + 1]), m);        // v_max3_f32
    };

    stage(smem, 0, ONLINE);
    __builtin_amdgcn_s_waitcnt(0x0f70);       // vmcnt(0)
    __syncthreads();

    // ---- pass 1: global row maximum of the raw scores.  Chunk c is computed out of buffer c & 1 while the DMA of chunk c + 1
    // fills the other one; after the last chunk comes chunk 0 of pass 2 (with V).
    float mxs = -INFINITY;
    for (int c = 0; c < nch && !ONLINE; ++c) {
        const int key0 = c * CK;
        const char *cur = smem + (c & 1) * BUF;
        char *nxt = smem + ((c + 1) & 1) * BUF;
        const bool last = c + 1 == nch;
        if (!(FLAGS & 1) || last) stage(nxt, last ? 0 : key0 + CK, last);
        if (wave_live && !(FLAGS & 16)) {
            const int nt = min(2, (N - key0 + 31) / 32);
            for (int kt = 0; kt < nt; ++kt) {
                f32x16 s; qk_tile(cur, kt, s);
                if (last) mask_tile(kt, key0, s);
                max_tile(s, mxs);
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);
        if (!(FLAGS & 2) || last) __syncthreads();
    }
    if constexpr (!ONLINE) mxs = fmaxf(mxs, __shfl_xor(mxs, 32));

    // ---- pass 2: exponentials against the global maximum (ONLINE: the running one), row sum of the ROUNDED values, O^T = V^T P^T
    float sum = 0.0f;
    float nmx = -AttnExp<T>::kScale * mxs;            // ONLINE: mxs = -inf here, nmx is set by the first tile's rescale
    f32x16 o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.0f;
    auto exp_tile = [&](const f32x16 &s, v8 (&p)[2]) {      // e = round(exp(round(s/8 - max/8))) per ggml_soft_max, row sum of the rounded values
        if constexpr (FLAGS & 4) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const typename Pair<T>::v2 eh = round_pair<T>(s[r], s[r + 1]);
                p[r >> 3][r & 7] = eh[0]; p[r >> 3][(r & 7) + 1] = eh[1];
            }
            sum += s[0];
        } else {
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const typename Pair<T>::v2 eh = AttnExp<T>::pair(s[r], s[r + 1], nmx);
                sum = Pair<T>::sum2(eh, sum);
                p[r >> 3][r & 7] = eh[0]; p[r >> 3][(r & 7) + 1] = eh[1];
            }
        }
    };
    // The V^T fragments are read with inline asm: behind the builtin hipcc puts `s_waitcnt vmcnt(0)` in front of every transposed
    // read (it cannot tell the read from the in-flight LDS-DMA of the NEXT chunk), which exposed the whole DMA latency per chunk.
    // "=v" results + one wait statement that owns them keeps the MFMAs below the wait.
    const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)((__attribute__((address_space(3))) char *)smem);
    auto pv_tile = [&](const char *cur, int kt, const v8 (&p)[2]) {      // O^T += V^T tile . P^T tile (4 MFMAs, 8 transposed reads)
        if constexpr (FLAGS & 8) { o[0][0] += (float)p[0][0] + (float)p[1][0]; } else {
            const unsigned cb = lds0 + (unsigned)(cur - smem) + kt * 32 * 128;
            s4 f[2][2][2];
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const unsigned va = cb + vrd[dt];
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f[dt][0][0]) : "v"(va));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(f[dt][0][1]) : "v"(va));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(f[dt][1][0]) : "v"(va));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:3072" : "=v"(f[dt][1][1]) : "v"(va));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0][0][0]), "+v"(f[0][0][1]), "+v"(f[0][1][0]), "+v"(f[0][1][1]),
                                                  "+v"(f[1][0][0]), "+v"(f[1][0][1]), "+v"(f[1][1][0]), "+v"(f[1][1][1]));
            typedef short s8 __attribute__((ext_vector_type(8)));
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const s8 both = __builtin_shufflevector(f[dt][half][0], f[dt][half][1], 0, 1, 2, 3, 4, 5, 6, 7);
                    o[dt] = Elem<T>::mfma(__builtin_bit_cast(v8, both), p[half], o[dt]);
                }
        }
    };
    const int buf0 = ONLINE ? 0 : nch;                // pass 1's last stage filled buffer nch & 1: pass 2's chunk c lives in buffer (nch + c) & 1
    for (int c = 0; c < nch; ++c) {
        const int key0 = c * CK;
        const char *cur = smem + ((buf0 + c) & 1) * BUF;
        char *nxt = smem + ((buf0 + c + 1) & 1) * BUF;
        const bool last = c + 1 == nch;
        if (!last && !(FLAGS & 1)) stage(nxt, key0 + CK, true);
        if (wave_live) {
            const int nt = min(2, (N - key0 + 31) / 32);
            for (int kt = 0; kt < nt; ++kt) {
                f32x16 s; v8 p[2];
                qk_tile(cur, kt, s);
                if (last) mask_tile(kt, key0, s);
                if constexpr (ONLINE) {
                    // a processed tile holds at least one real key, so its maximum is finite; the two lanes of a query (key halves hh = 0, 1) see
                    // the same tile maximum and the same running one, hence the same shift
                    constexpr float kTau = 8.0f / AttnExp<T>::kScale;
                    float tm = -INFINITY, u, v;
                    max_tile(s, tm);
                    rows_swap32(tm, u, v); tm = fmaxf(u, v);
                    if (__builtin_amdgcn_ballot_w64(tm > mxs + kTau) != 0) {          // wave-uniform; first tile: mxs = -inf
                        const float mnew = fmaxf(mxs, tm);
                        const float sc = __builtin_amdgcn_exp2f((mxs - mnew) * AttnExp<T>::kScale);      // exp2(-inf) = 0 on the first tile
                        mxs = mnew; nmx = -AttnExp<T>::kScale * mnew;
                        sum *= sc;
#pragma unroll
                        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                            for (int r = 0; r < 16; ++r) o[dt][r] *= sc;
                    }
                }
                exp_tile(s, p); pv_tile(cur, kt, p);
            }
        }
        if (!last) { __builtin_amdgcn_s_waitcnt(0x0f70); if (!(FLAGS & 2)) __syncthreads(); }
    }
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
    if (qvalid) {
        T *orow = out + ((size_t)b * N + qrow) * D + h * 64;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                typename Elem<T>::v4 w4;
#pragma unroll
                for (int e = 0; e < 4; ++e) w4[e] = (T)(o[dt][r4 * 4 + e] * inv);
                *(typename Elem<T>::v4 *)(orow + dt * 32 + r4 * 8 + hh * 4) = w4;
            }
    }
}
template <typename T, int FLAGS>
static hipError_t launch_attention_flow_inst(const void *qkv, void *out, int n_img, int N, int D, int H, hipStream_t stream) {
    constexpr int lds = 2 * (64 * 128 + 64 * 128);        // two (8 KiB K + 8 KiB V) chunk buffers
    constexpr bool ONLINE = std::is_same<T, __bf16>::value && FLAGS == 0;
    if (n_img == 0) return hipFuncSetAttribute((const void *)attention_flow_kernel<T, FLAGS, ONLINE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);   // device bring-up
    const int qblocks = ((N + 31) / 32 + 3) / 4, items = n_img * H;
    const int grid = ((items + 7) / 8) * 8 * qblocks;
    hipLaunchKernelGGL((attention_flow_kernel<T, FLAGS, ONLINE>), dim3(grid), dim3(256), lds, stream, (const T *)qkv, (T *)out, N, D, H, qblocks, items, n_img);
    return hipGetLastError();
}
template <typename T>
static hipError_t launch_attention_flow(const void *qkv, void *out, int n_img, int N, int D, int H, hipStream_t stream, int flags = 0) {
    switch (flags) {
    case 0: return launch_attention_flow_inst<T, 0>(qkv, out, n_img, N, D, H, stream);
#ifdef VITX_LAB      // ablation builds of tools/attn_bench.py (results are garbage by design)
    case 1: return launch_attention_flow_inst<T, 1>(qkv, out, n_img, N, D, H, stream);
    case 2: return launch_attention_flow_inst<T, 2>(qkv, out, n_img, N, D, H, stream);
    case 4: return launch_attention_flow_inst<T, 4>(qkv, out, n_img, N, D, H, stream);
    case 8: return launch_attention_flow_inst<T, 8>(qkv, out, n_img, N, D, H, stream);
    case 16: return launch_attention_flow_inst<T, 16>(qkv, out, n_img, N, D, H, stream);
#endif
    default: return hipErrorInvalidValue;
    }
}

// ------------------------------------------------------------------------------------------------
#ifndef ATT_AUX
#define ATT_AUX 2               // cache-policy bits of the persistent attention kernel's K / V LDS-DMA: nt -- QKV is read once (gemm_pp.hip "Cache-policy bits", profiles/r05/ab_cache_policy.txt)
#endif
#ifndef ATT_OUT_AUX
#define ATT_OUT_AUX 0           // cache-policy bits of the persistent attention kernel's output stores (read once, by proj)
#endif
#ifndef PERSIST_SUM_MFMA
#define PERSIST_SUM_MFMA 1
#endif
// Persistent single-pass attention (vit.cpp:826-866) for 193..224 tokens -- ViT-*/16 at 224^2, the headline configuration.
//   * one persistent workgroup per CU, SIXTEEN waves (four per SIMD), walks (image, head) items; wave w owns the 16 queries 16 w .. 16 w + 15
//     of the item (197 tokens: 13 waves compute, three only move data).  A wave issues its softmax arithmetic in order, about one
//     instruction per 6 cycles (tools/issue_probe.hip: 32 v_fmamk 196 cycles, 16 v_exp_f32 168, for one wave and for two), and a SIMD
//     arbitrates by age: with eight waves of 32 queries the second wave of a SIMD took 5500 cycles for the softmax the first one did in
//     3300 (clock stamps, profiles/r03/attention_stamps.txt);
//   * K (swizzled row image, permutation on the source side) and V (row-major, 32-byte chunks XOR-ed with (row >> 1) & 3) land by LDS-DMA
//     -- no staging registers, no VALU, no LDS transposition -- in a ring of item buffers: THREE for 193..208 tokens (208-row images,
//     3 x 52 KiB = 156 of the 160 KiB), so item i + 2 is requested while item i is computed and the memory system always holds one whole
//     item per CU beyond the one being waited for (with two buffers the launch ran 36 us where its memory traffic alone takes 25 and its
//     arithmetic alone 27); two buffers of 224 rows above 208 tokens;
//   * products are v_mfma_f32_16x16x32 (the GEMMs' instruction: 11 % less energy per flop than 32x32x16 on this part):
//     S^T = K . Q^T as 16-key x 16-query tiles, so a query's scores sit in the 4 lanes (lane & 15, lane >> 4 = 0..3) and the softmax
//     reductions are in-register plus two cross-row shuffles; the probabilities go from the accumulator registers straight into the
//     B operand of O^T = V^T . P^T (k-slot j of lane group g = key 4 g + j of the first, 16 + 4 g + (j - 4) of the second 16-key tile
//     of a 32-key step), and the V^T fragments in that same key order come out of two ds_read_b64_tr_b16 each; an odd last key tile is
//     one v_mfma_f32_16x16x16 (same operand layout, half the k-slots);
//   * key tiles and query tiles that hold no real token do not exist (NT16V: compile-time), so an item's work is ONE basic block;
//     every LDS read is inline asm with counted lgkmcnt, K fragments two tiles and V^T fragments one key step ahead of their products
//     (behind the builtins hipcc drains the DMA queue, vmcnt(0), in front of each read -- any of them might alias a landing piece);
//   * the output leaves as 16-byte stores of whole 64-byte lines: v_permlane16_swap trades the odd head-dim tile of the even lane
//     row for the even tile of the odd row, so a lane holds 8 consecutive dims of its query (half the store instructions);
//   * one barrier per item; vector-memory operations retire in issue order, so the wait in front of it counts what was issued AFTER
//     the loads it needs: the DMA pieces of item i + 2 and this item's stores stay in flight.
// exp follows AttnExp<T> (device_common.h): F16 = ggml_soft_max's table semantics, BF16 = one f32 exp2 per key.
// Keys N .. 16 NT16V - 1 read the next image's rows (finite; masked to -inf) or the zeros a buffer load returns out of range.
// ------------------------------------------------------------------------------------------------
// QT = 1 for a wave that computes, 0 for one that only moves data; NT16V = 16-key tiles that hold a real key (13 for 193..208 tokens,
// 14 above).  FLAGS: 0 in the product; ablation builds under -DVITX_LAB only (tools/attn_bench.py, garbage results by design): 1 = no
// softmax arithmetic, 2 = no K / V DMA after the first items, 4 = no output stores, 8 = no QK^T products, 16 = no PV products, 32 = no Q
// loads, 64 = shader-clock stamps per phase behind the output rows.
template <typename T, int NT16V, int QT, int FLAGS>
__device__ __forceinline__ void attention_persist_loop(const T *__restrict__ qkv, T *__restrict__ out, char *smem, int N, int D, int H, int items, unsigned total_bytes, unsigned out_bytes) {
    constexpr int NROW = NT16V * 16, KB = NROW * 128, BUF = 2 * KB;    // one item: K image + V image
    constexpr int NBUF = NT16V <= 13 ? 3 : 2, AHEAD = NBUF - 1;         // ring of item buffers, items requested ahead
    constexpr int NTHR = 1024, PIECES = NROW * 8, OPS = (PIECES + NTHR - 1) / NTHR;   // 16-byte pieces per image, DMA instructions per thread and image
    constexpr int NKT = (NT16V + 1) / 2;                                // 32-key steps of the PV product (the last one may be half)
    static_assert(OPS == 2 && NKT == 7, "193..224 tokens");
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g4 = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    typedef typename Elem<T>::v8 v8;
    typedef typename Pair<T>::v2 v2;
    typedef short s4 __attribute__((ext_vector_type(4)));
    typedef short s8 __attribute__((ext_vector_type(8)));
    typedef int i4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    const int row_bytes = 3 * D * 2;
    const bool second = NTHR + wave * 64 < PIECES;      // wave-uniform: this wave also issues the second (partial) DMA instruction of an image

    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)qkv, 0, (int)total_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rsrc_o = __builtin_amdgcn_make_buffer_rsrc((void *)out, 0, (int)out_bytes, 0x00020000);
    // bytes (< 4 GiB: launcher) -- UNSIGNED: as an int it went negative beyond 2 GiB and load_q's pointer arithmetic sign-extended it (r04: wrong
    // results from 2366 ViT-B images per launch on, a fault beyond; found by the chunked-launch test)
    auto item_base = [&](int item) -> unsigned { const int b = item / H, h = item - b * H; return (unsigned)(((size_t)b * N * 3 * D + h * 64) * 2); };
    // DMA piece it * 1024 + tid of an image is image row (128 it + row of piece tid), same 16-byte slot: ONE per-lane offset per image
    // and an SGPR stride
    int koff0, voff0;
    {
        int rr, sl; swz_inv(tid & 511, rr, sl);                                    // 512 pieces = one 64-row block of the swizzled image
        koff0 = ((tid >> 9) * 64 + rr) * row_bytes + D * 2 + sl * 16;
        const int vr = tid >> 3, vs = (tid & 7) ^ (((vr >> 1) & 3) << 1);          // V image: 32-byte chunk ^ ((row >> 1) & 3); rows 128 it + vr share it
        voff0 = vr * row_bytes + 2 * D * 2 + vs * 16;
    }
    auto stage = [&](int item, char *buf) {
        const int so = __builtin_amdgcn_readfirstlane((int)item_base(item));          // the buffer unit takes the SGPR offset as 32 unsigned bits
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(buf + wave * 1024), 16, koff0, so, 0, ATT_AUX);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(buf + KB + wave * 1024), 16, voff0, so, 0, ATT_AUX);
        if (second) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(buf + NTHR * 16 + wave * 1024), 16, koff0, so + (NTHR / 8) * row_bytes, 0, ATT_AUX);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(buf + KB + NTHR * 16 + wave * 1024), 16, voff0, so + (NTHR / 8) * row_bytes, 0, ATT_AUX);
        }
    };
    // Q fragments (B operand of S^T = K . Q^T): lane (l15 = query of the tile, g4) holds dims k2 * 32 + g4 * 8 .. + 7
    v8 qf[2];
    auto load_q = [&](int item) {
        const T *base = qkv + (size_t)item_base(item) / 2;
        const int qrow = min(wave * 16 + l15, N - 1);
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) qf[k2] = *(const v8 *)(base + (size_t)qrow * 3 * D + k2 * 32 + g4 * 8);
    };
    // fragment addresses: K tile t (16 keys) = parity (t & 1) base + (t >> 1) * 4096; V step ks (32 keys) adds ks * 4096, its second half 2048
    int krd[2][2], vrd[4];
#pragma unroll
    for (int pz = 0; pz < 2; ++pz)
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) krd[pz][k2] = swz_byte(pz * 16 + l15, k2 * 4 + g4);
    {
        const int r = 4 * g4 + (l15 >> 2), x = (r >> 1) & 3;       // this lane's V row within a 16-key group and its chunk swizzle
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) vrd[dt] = KB + r * 128 + ((dt ^ x) << 5) + (l15 & 3) * 8;
    }
    const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)((__attribute__((address_space(3))) char *)smem);
    // output: after the row swap a lane holds dims 8 (g4 >> 1) .. + 7 of head-dim tile 2 pr + (g4 & 1), pr = 0, 1
    const int st_lane = (g4 & 1) * 32 + (g4 >> 1) * 16;

    int item = blockIdx.x;
    int cur_off = 0;
    stage(item, smem);
    if (AHEAD == 2 && item + (int)gridDim.x < items) stage(item + gridDim.x, smem + BUF);
    if constexpr (QT > 0) load_q(item);
    __builtin_amdgcn_s_waitcnt(0x0f70);       // vmcnt(0)
    __builtin_amdgcn_s_barrier();

    for (; item < items; item += gridDim.x) {
        const int b = item / H, h = item - b * H;
        const int nitem = item + gridDim.x, aitem = item + AHEAD * gridDim.x;      // the next item; the item requested during this one
        const bool ahead = aitem < items && !(FLAGS & 2);
        const int aoff = cur_off + AHEAD * BUF >= NBUF * BUF ? cur_off + AHEAD * BUF - NBUF * BUF : cur_off + AHEAD * BUF;
        f32x4 s[14];
        unsigned long long stamp[6];
#define VITX_STAMP(I) if (FLAGS & 64) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp[I]) :: "memory");
        VITX_STAMP(0)
        if constexpr (QT > 0) {
            // S^T tiles: rows = keys, cols = queries.  The K fragments of tile t + KD are requested before the products of tile t
            constexpr int KD = 2, KS = KD + 1;
            i4 kf[KS][2];
            unsigned ka[2][2];
#pragma unroll
            for (int pz = 0; pz < 2; ++pz)
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) ka[pz][k2] = lds0 + (unsigned)cur_off + krd[pz][k2];
            auto read_k = [&](int t) {
                const int sl = t % KS;
                switch (t >> 1) {       // the immediate offset must be a literal
#define VITX_RK(I) case I: asm volatile("ds_read_b128 %0, %1 offset:" #I "*4096" : "=v"(kf[sl][0]) : "v"(ka[t & 1][0])); asm volatile("ds_read_b128 %0, %1 offset:" #I "*4096" : "=v"(kf[sl][1]) : "v"(ka[t & 1][1])); break;
                VITX_RK(0) VITX_RK(1) VITX_RK(2) VITX_RK(3) VITX_RK(4) VITX_RK(5) VITX_RK(6)
#undef VITX_RK
                }
            };
#pragma unroll
            for (int t = 0; t < KD; ++t) read_k(t);
#pragma unroll
            for (int t = 0; t < NT16V; ++t) {
                const int sl = t % KS;
                if (t + KD < NT16V) read_k(t + KD);
                const int behind = 2 * ((t + KD < NT16V ? t + KD : NT16V - 1) - t);        // reads requested after tile t's
                switch (behind) {
#define VITX_WK(C) case C: asm volatile("s_waitcnt lgkmcnt(" #C ")" : "+v"(kf[sl][0]), "+v"(kf[sl][1])); break;
                VITX_WK(0) VITX_WK(2) VITX_WK(4)
#undef VITX_WK
                }
                const v8 k0 = __builtin_bit_cast(v8, kf[sl][0]), k1 = __builtin_bit_cast(v8, kf[sl][1]);
                if (FLAGS & 8) { s[t] = f32x4{k0[0], k1[1], k0[2], k1[3]}; continue; }
                f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
                acc = Elem<T>::mfma16(k0, qf[0], acc);
                s[t] = Elem<T>::mfma16(k1, qf[1], acc);
            }
        }
        VITX_STAMP(1)
        __builtin_amdgcn_sched_barrier(0);
        // Loads for later items, oldest need first (they retire in this order): the next item's Q fragments (this item's are dead), then
        // the K / V pieces of the item AHEAD, into the buffer the whole workgroup left at the last barrier.  Issued here and not at the
        // top of the item: beside the K-fragment reads of every wave the DMA cost 1.8 us per 128-image launch.
        if constexpr (QT > 0) if (nitem < items && !(FLAGS & 32)) load_q(nitem);
        if (ahead) stage(aitem, smem + aoff);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (QT > 0) {
            v8 p[7];
            float inv;
            if (FLAGS & 1) {
#pragma unroll
                for (int ks = 0; ks < NKT; ++ks) {
                    const f32x4 a = s[2 * ks], c = s[2 * ks + 1 < NT16V ? 2 * ks + 1 : 0];
                    const v2 e0 = round_pair<T>(a[0], a[1]), e1 = round_pair<T>(a[2], a[3]), e2 = round_pair<T>(c[0], c[1]), e3 = round_pair<T>(c[2], c[3]);
                    p[ks] = v8{e0[0], e0[1], e1[0], e1[1], e2[0], e2[1], e3[0], e3[1]};
                }
                inv = 1.0f;
            } else {
                // row maximum: four independent chains (a single one is 26 dependent v_max3), rows combined in the VALU
                float mx4[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
                for (int t = 0; t < NT16V; ++t) {
                    if (t >= 12) {       // only the last two 16-key tiles can hold padded keys (N > 192)
#pragma unroll
                        for (int r = 0; r < 4; ++r) if (t * 16 + 4 * g4 + r >= N) s[t][r] = -INFINITY;
                    }
                    mx4[t & 3] = fmaxf(fmaxf(mx4[t & 3], s[t][0]), s[t][1]);       // v_max3_f32
                    mx4[t & 3] = fmaxf(fmaxf(mx4[t & 3], s[t][2]), s[t][3]);
                }
                const float mxs = rows4_max(fmaxf(fmaxf(mx4[0], mx4[1]), fmaxf(mx4[2], mx4[3])));
                const float nmx = -AttnExp<T>::kScale * mxs;
                float sum2[2] = {0.0f, 0.0f};
#pragma unroll
                for (int ks = 0; ks < NKT; ++ks) {      // numerators per AttnExp<T>; row sum of the ROUNDED values (they are what the PV product sees)
                    const v2 e0 = AttnExp<T>::pair(s[2 * ks][0], s[2 * ks][1], nmx), e1 = AttnExp<T>::pair(s[2 * ks][2], s[2 * ks][3], nmx);
                    if (!PERSIST_SUM_MFMA) { sum2[0] = Pair<T>::sum2(e0, sum2[0]); sum2[0] = Pair<T>::sum2(e1, sum2[0]); }
                    v2 e2 = __builtin_bit_cast(v2, 0u), e3 = __builtin_bit_cast(v2, 0u);       // (an odd last tile: these slots are not multiplied)
                    if (2 * ks + 1 < NT16V) {
                        e2 = AttnExp<T>::pair(s[2 * ks + 1][0], s[2 * ks + 1][1], nmx); e3 = AttnExp<T>::pair(s[2 * ks + 1][2], s[2 * ks + 1][3], nmx);
                        if (!PERSIST_SUM_MFMA) { sum2[1] = Pair<T>::sum2(e2, sum2[1]); sum2[1] = Pair<T>::sum2(e3, sum2[1]); }
                    }
                    p[ks] = v8{e0[0], e0[1], e1[0], e1[1], e2[0], e2[1], e3[0], e3[1]};
                }
                if (!PERSIST_SUM_MFMA) { const float sum = rows4_sum(sum2[0] + sum2[1]); inv = 1.0f / sum; }
            }
            VITX_STAMP(2)
            // O^T = V^T . P^T: rows = head dims (4 tiles of 16), cols = queries; V^T fragments by transposed LDS reads, the reads of key
            // step ks + 1 issued ahead of the products of step ks (LDS operations return in order: lgkmcnt(n) = all but the youngest n landed)
            f32x4 o[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            // PERSIST_SUM_MFMA: the row sums come out of the matrix pipe -- a fifth "head-dim tile" whose V^T rows are all ones gives every lane the sum of
            // ITS query's rounded numerators (exact products with 1.0, f32 accumulation) -- instead of 26 v_dot2c per wave and item plus the cross-lane sum
            f32x4 osum = {0.0f, 0.0f, 0.0f, 0.0f};
            const unsigned one2 = std::is_same<T, __bf16>::value ? 0x3f803f80u : 0x3c003c00u;
            const s8 ones8 = __builtin_bit_cast(s8, (u32x4_t{one2, one2, one2, one2}));
            s4 f[2][4][2];
            constexpr bool HALF = (NT16V & 1) != 0;         // the last key step holds one 16-key tile
            auto read_v = [&](int ks) {
                const unsigned cb = lds0 + (unsigned)cur_off + ks * 4096;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const unsigned va = cb + vrd[dt];
                    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f[ks & 1][dt][0]) : "v"(va));
                    if (!(HALF && ks == NKT - 1)) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(f[ks & 1][dt][1]) : "v"(va));
                }
            };
            read_v(0);
#pragma unroll
            for (int ks = 0; ks < NKT; ++ks) {
                const int c = ks & 1;
                if (ks + 1 < NKT) {
                    read_v(ks + 1);
                    if (HALF && ks + 1 == NKT - 1)
                        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(f[c][0][0]), "+v"(f[c][0][1]), "+v"(f[c][1][0]), "+v"(f[c][1][1]),
                                                              "+v"(f[c][2][0]), "+v"(f[c][2][1]), "+v"(f[c][3][0]), "+v"(f[c][3][1]));
                    else
                        asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(f[c][0][0]), "+v"(f[c][0][1]), "+v"(f[c][1][0]), "+v"(f[c][1][1]),
                                                              "+v"(f[c][2][0]), "+v"(f[c][2][1]), "+v"(f[c][3][0]), "+v"(f[c][3][1]));
                } else if (HALF) {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[c][0][0]), "+v"(f[c][1][0]), "+v"(f[c][2][0]), "+v"(f[c][3][0]));
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[c][0][0]), "+v"(f[c][0][1]), "+v"(f[c][1][0]), "+v"(f[c][1][1]),
                                                          "+v"(f[c][2][0]), "+v"(f[c][2][1]), "+v"(f[c][3][0]), "+v"(f[c][3][1]));
                }
                if (PERSIST_SUM_MFMA && !(FLAGS & 1)) {
                    if (HALF && ks == NKT - 1) { const s8 pk = __builtin_bit_cast(s8, p[ks]); osum = Elem<T>::mfma16k16(s4{ones8[0], ones8[1], ones8[2], ones8[3]}, s4{pk[0], pk[1], pk[2], pk[3]}, osum); }
                    else osum = Elem<T>::mfma16(__builtin_bit_cast(v8, ones8), p[ks], osum);
                }
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    if (HALF && ks == NKT - 1) {
                        const s8 pk = __builtin_bit_cast(s8, p[ks]);
                        if (FLAGS & 16) { o[dt][0] += (float)f[c][dt][0][0] + (float)pk[dt]; continue; }
                        o[dt] = Elem<T>::mfma16k16(f[c][dt][0], s4{pk[0], pk[1], pk[2], pk[3]}, o[dt]);
                        continue;
                    }
                    const s8 both = __builtin_shufflevector(f[c][dt][0], f[c][dt][1], 0, 1, 2, 3, 4, 5, 6, 7);
                    if (FLAGS & 16) { o[dt][0] += (float)both[0] + (float)p[ks][dt]; continue; }
                    o[dt] = Elem<T>::mfma16(__builtin_bit_cast(v8, both), p[ks], o[dt]);
                }
            }
            if (PERSIST_SUM_MFMA && !(FLAGS & 1)) inv = 1.0f / osum[0];            // every row of the ones tile holds the query's sum
            VITX_STAMP(3)
            // lane (l15 = query, g4) holds O[query][dt * 16 + 4 g4 .. + 3].  v_permlane16_swap: the even lane row gives its odd tile and
            // takes the odd row's even tile -> 8 consecutive dims per lane, two 16-byte stores per wave, each covering whole 64-byte
            // lines; rows past N go out of the buffer's range and are dropped, so a wave issues exactly 2 stores per item (the counted
            // wait below relies on it)
            const int qrow = wave * 16 + l15;
            const unsigned off = qrow < N ? (unsigned)((((size_t)b * N + qrow) * D + h * 64) * 2 + st_lane) : 0xffffff00u;
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                const f32x4 oe = o[2 * pr], oo = o[2 * pr + 1];
                const v2 elo = round_pair<T>(oe[0] * inv, oe[1] * inv), ehi = round_pair<T>(oe[2] * inv, oe[3] * inv);
                const v2 olo = round_pair<T>(oo[0] * inv, oo[1] * inv), ohi = round_pair<T>(oo[2] * inv, oo[3] * inv);
                const auto lo = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, elo), __builtin_bit_cast(unsigned, olo), false, false);
                const auto hi = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, ehi), __builtin_bit_cast(unsigned, ohi), false, false);
                if (FLAGS & 4) { if (lo[0] == 0x12345678u && hi[1] == 0x9abcdef0u) out[off] = (T)1.0f; continue; }       // keeps the values alive, never true
                __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{lo[0], hi[0], lo[1], hi[1]}, rsrc_o, (int)(off + pr * 64), 0, ATT_OUT_AUX);
            }
        }
        VITX_STAMP(4)
        // The next item's K / V (requested AHEAD items ago, or just now with two buffers) and Q must have landed.  Younger than those loads
        // and allowed to stay in flight: with three buffers the pieces of item i + 2 (2 or 4 per wave), and this wave's 2 output stores.
        {
            constexpr int ST = (QT > 0 && !(FLAGS & 4)) ? 2 : 0;
            if (AHEAD == 2 && ahead) {
                if (second) { if constexpr (ST) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
                else { if constexpr (ST) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
            } else {
                if constexpr (ST) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        VITX_STAMP(5)
        if ((FLAGS & 64) && blockIdx.x == 0 && lane == 0 && item / (int)gridDim.x < 6) {       // lab: [wave][item][6] shader-clock stamps behind the output rows
            unsigned long long *dbg = (unsigned long long *)((char *)out + out_bytes) + (wave * 6 + item / gridDim.x) * 6;
            for (int i = 0; i < 6; ++i) dbg[i] = (QT == 0 && i >= 2 && i <= 3) ? stamp[1] : stamp[i];
        }
#undef VITX_STAMP
        __builtin_amdgcn_s_barrier();             // every wave is done with this item's buffer; the next one is visible to all
        if (!(FLAGS & 2)) cur_off = cur_off + BUF >= NBUF * BUF ? 0 : cur_off + BUF;
    }
}

template <typename T, int NT16V, int FLAGS = 0>
__global__ __launch_bounds__(1024) void attention_persist_kernel(const T *__restrict__ qkv, T *__restrict__ out, int N, int D, int H, int items, unsigned total_bytes, unsigned out_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((int)blockIdx.x >= items) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // every wave runs the same number of items and barriers, whichever build of the loop it takes
    if (wave < NT16V) attention_persist_loop<T, NT16V, 1, FLAGS>(qkv, out, smem, N, D, H, items, total_bytes, out_bytes);
    else attention_persist_loop<T, NT16V, 0, FLAGS>(qkv, out, smem, N, D, H, items, total_bytes, out_bytes);
}
template <int NT16V> constexpr int attention_persist_lds() { return (NT16V <= 13 ? 3 : 2) * 2 * NT16V * 16 * 128; }      // 156 KiB / 112 KiB
template <typename T>
static hipError_t launch_attention_persist(const void *qkv, void *out, int n_img, int N, int D, int H, int n_cu, hipStream_t stream, int flags = 0) {
    if (n_img == 0) {       // device bring-up
        hipError_t e = hipFuncSetAttribute((const void *)attention_persist_kernel<T, 13>, hipFuncAttributeMaxDynamicSharedMemorySize, attention_persist_lds<13>());
        return e != hipSuccess ? e : hipFuncSetAttribute((const void *)attention_persist_kernel<T, 14>, hipFuncAttributeMaxDynamicSharedMemorySize, attention_persist_lds<14>());
    }
    const size_t total = (size_t)n_img * N * 3 * D * 2;
    if (total >= 0xf0000000u) return hipErrorInvalidValue;          // 32-bit buffer offsets
    const int items = n_img * H;
    // balanced persistent grid (as the GEMMs'): every workgroup walks the same number of items; the CUs a partial last round would have
    // lit for one item go to the other sub-batch's kernels for the whole launch (1836 items: 230 workgroups x 8 instead of 256 x 7.2)
    const int rounds = (items + n_cu - 1) / n_cu;
    const int grid = (items + rounds - 1) / rounds;
#ifdef VITX_LAB
#define VITX_PERSIST_LAB(F) case F: { static bool once = false; if (!once) { once = true; (void)hipFuncSetAttribute((const void *)attention_persist_kernel<T, 13, F>, hipFuncAttributeMaxDynamicSharedMemorySize, attention_persist_lds<13>()); } \
        hipLaunchKernelGGL((attention_persist_kernel<T, 13, F>), dim3(grid), dim3(1024), attention_persist_lds<13>(), stream, (const T *)qkv, (T *)out, N, D, H, items, (unsigned)total, (unsigned)(total / 3)); return hipGetLastError(); }
    if (flags && N <= 208) switch (flags) {
        VITX_PERSIST_LAB(1) VITX_PERSIST_LAB(2) VITX_PERSIST_LAB(4) VITX_PERSIST_LAB(8) VITX_PERSIST_LAB(16) VITX_PERSIST_LAB(32) VITX_PERSIST_LAB(25) VITX_PERSIST_LAB(38) VITX_PERSIST_LAB(64)
        default: return hipErrorInvalidValue; }
#undef VITX_PERSIST_LAB
#endif
    if (N <= 208) hipLaunchKernelGGL((attention_persist_kernel<T, 13>), dim3(grid), dim3(1024), attention_persist_lds<13>(), stream, (const T *)qkv, (T *)out, N, D, H, items, (unsigned)total, (unsigned)(total / 3));
    else hipLaunchKernelGGL((attention_persist_kernel<T, 14>), dim3(grid), dim3(1024), attention_persist_lds<14>(), stream, (const T *)qkv, (T *)out, N, D, H, items, (unsigned)total, (unsigned)(total / 3));
    return hipGetLastError();
}
// ------------------------------------------------------------------------------------------------
// Attention for head dimensions other than 64 (vit.cpp:826-866 is generic in n_enc_head_dim; timm's ViT-H/14 has 80, 8-head variants
// 96 / 128, small models 32): any multiple of 8 up to 128.  Not a tuned kernel -- every model the benchmarks name has head_dim 64 -- but
// the same arithmetic as the other three: S^T = K . Q^T by v_mfma_f32_16x16x32 over the head dim zero-padded to a multiple of 32, two
// passes over the keys (row maximum; then exp per AttnExpRt<T>, row sum of the rounded numerators, O^T = V^T . P^T), scores in the
// four lanes (lane & 15, lane >> 4) of a query.  One wave per 16-query tile, four tiles per workgroup, no workgroup barrier: K
// fragments come straight from global memory (16-byte pieces of a key's head slice; pieces past head_dim are zeros), a 32-key step
// of V goes through the wave's own 8 KiB of LDS to be read back transposed (ds_read_b64_tr_b16).
// ------------------------------------------------------------------------------------------------
template <typename T, int NK2>
__global__ __launch_bounds__(256) void attention_generic_kernel(const T *__restrict__ qkv, T *__restrict__ out, int N, int D, int H, int DH, float scale, int qblocks) {
    constexpr int DHP = NK2 * 32, ND = DHP / 16, ROWB = DHP * 2;       // padded head dim, 16-dim output tiles, LDS row bytes
    __shared__ __attribute__((aligned(16))) char smem[4 * 32 * ROWB];
    typedef typename Elem<T>::v8 v8;
    typedef typename Pair<T>::v2 v2;
    typedef short s4 __attribute__((ext_vector_type(4)));
    typedef short s8 __attribute__((ext_vector_type(8)));
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g4 = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int item = blockIdx.x / qblocks, qb = blockIdx.x - item * qblocks;
    const int b = item / H, h = item - b * H;
    const int q0 = (qb * 4 + wave) * 16;
    if (q0 >= N) return;                              // no barrier in this kernel: a wave without queries simply leaves
    const size_t row_el = (size_t)3 * D;
    const T *base = qkv + (size_t)b * N * row_el + (size_t)h * DH;        // q of token 0; k at + D, v at + 2 D
    const v8 zero8 = __builtin_bit_cast(v8, (int __attribute__((ext_vector_type(4)))){0, 0, 0, 0});
    // Q fragments (B operand): lane (l15 = query, g4) holds dims k2 * 32 + g4 * 8 .. + 7
    v8 qf[NK2];
    {
        const int qrow = min(q0 + l15, N - 1);
#pragma unroll
        for (int k2 = 0; k2 < NK2; ++k2) { const int d0 = k2 * 32 + g4 * 8; qf[k2] = d0 < DH ? *(const v8 *)(base + (size_t)qrow * row_el + d0) : zero8; }
    }
    auto score_tile = [&](int t) {                   // S^T tile t: rows = keys 16 t .., cols = queries; acc[r] = key 16 t + 4 g4 + r
        const int krow = min(t * 16 + l15, N - 1);
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int k2 = 0; k2 < NK2; ++k2) {
            const int d0 = k2 * 32 + g4 * 8;
            const v8 kf = d0 < DH ? *(const v8 *)(base + D + (size_t)krow * row_el + d0) : zero8;
            acc = Elem<T>::mfma16(kf, qf[k2], acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) if (t * 16 + 4 * g4 + r >= N) acc[r] = -INFINITY;
        return acc;
    };
    // ---- pass 1: row maximum
    const int nt16 = (N + 15) / 16, nks = (N + 31) / 32;
    float mx = -INFINITY;
    for (int t = 0; t < nt16; ++t) {
        const f32x4 sc = score_tile(t);
        mx = fmaxf(fmaxf(mx, sc[0]), sc[1]); mx = fmaxf(fmaxf(mx, sc[2]), sc[3]);
    }
    mx = rows4_max(mx);
    const float kk = AttnExpRt<T>::k(scale), nmx = -kk * mx;
    // ---- pass 2: numerators, row sum, PV
    f32x4 o[ND];
#pragma unroll
    for (int dt = 0; dt < ND; ++dt) o[dt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    float sum = 0.0f;
    char *my = smem + wave * (32 * ROWB);
    const unsigned lds_my = (unsigned)(__UINTPTR_TYPE__)((__attribute__((address_space(3))) char *)my);
    const unsigned tr_off = (4 * g4 + (l15 >> 2)) * ROWB + (l15 & 3) * 8;      // this lane's V row of a 16-key group, 4-dim piece of a 16-dim tile
    for (int ks = 0; ks < nks; ++ks) {
        const f32x4 sa = score_tile(2 * ks);
        f32x4 sb = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        if (2 * ks + 1 < nt16) sb = score_tile(2 * ks + 1);
        const v2 e0 = AttnExpRt<T>::pair(sa[0], sa[1], nmx, kk), e1 = AttnExpRt<T>::pair(sa[2], sa[3], nmx, kk);
        const v2 e2 = AttnExpRt<T>::pair(sb[0], sb[1], nmx, kk), e3 = AttnExpRt<T>::pair(sb[2], sb[3], nmx, kk);
        sum = Pair<T>::sum2(e0, sum); sum = Pair<T>::sum2(e1, sum); sum = Pair<T>::sum2(e2, sum); sum = Pair<T>::sum2(e3, sum);
        const v8 pk = v8{e0[0], e0[1], e1[0], e1[1], e2[0], e2[1], e3[0], e3[1]};
        // V rows 32 ks .. + 31 (clamped: their probabilities are zero past N) x DHP dims into the wave's LDS, row-major
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the previous step's transposed reads are done with it
#pragma unroll
        for (int i = 0; i < (32 * DHP / 8 + 63) / 64; ++i) {
            const int pi = i * 64 + lane, row = pi / (DHP / 8), c8 = pi - row * (DHP / 8);
            if (pi < 32 * DHP / 8) {
                const int vrow = min(ks * 32 + row, N - 1);
                const v8 vv = c8 * 8 < DH ? *(const v8 *)(base + 2 * D + (size_t)vrow * row_el + c8 * 8) : zero8;
                *(v8 *)(my + row * ROWB + c8 * 16) = vv;
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int dt = 0; dt < ND; ++dt) {
            s4 f0, f1;
            const unsigned va = lds_my + tr_off + dt * 32;
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f0) : "v"(va) : "memory");
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f1) : "v"(va + 16 * ROWB) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f0), "+v"(f1));
            const s8 both = __builtin_shufflevector(f0, f1, 0, 1, 2, 3, 4, 5, 6, 7);
            o[dt] = Elem<T>::mfma16(__builtin_bit_cast(v8, both), pk, o[dt]);
        }
    }
    const float inv = 1.0f / rows4_sum(sum);
    // lane (l15 = query, g4) holds O[query][dt * 16 + 4 g4 .. + 3]
    const int qrow = q0 + l15;
    if (qrow < N) {
        T *orow = out + ((size_t)b * N + qrow) * D + (size_t)h * DH;
#pragma unroll
        for (int dt = 0; dt < ND; ++dt) {
            const int d0 = dt * 16 + 4 * g4;
            if (d0 < DH) {
                const v2 lo = round_pair<T>(o[dt][0] * inv, o[dt][1] * inv), hi = round_pair<T>(o[dt][2] * inv, o[dt][3] * inv);
                typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
                *(u32x2_t *)(orow + d0) = u32x2_t{__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
            }
        }
    }
}
bool attention_generic_supports(int D, int H) { return H > 0 && D % H == 0 && (D / H) % 8 == 0 && D / H >= 8 && D / H <= 128; }
template <typename T>
static hipError_t launch_attention_generic(const void *qkv, void *out, int n_img, int N, int D, int H, hipStream_t stream) {
    const int DH = D / H, nk2 = (DH + 31) / 32, qblocks = (N + 63) / 64;
    const float scale = 1.0f / sqrtf((float)DH);
    const dim3 grid((unsigned)((size_t)n_img * H * qblocks)), blk(256);
    switch (nk2) {
    case 1: hipLaunchKernelGGL((attention_generic_kernel<T, 1>), grid, blk, 0, stream, (const T *)qkv, (T *)out, N, D, H, DH, scale, qblocks); break;
    case 2: hipLaunchKernelGGL((attention_generic_kernel<T, 2>), grid, blk, 0, stream, (const T *)qkv, (T *)out, N, D, H, DH, scale, qblocks); break;
    case 3: hipLaunchKernelGGL((attention_generic_kernel<T, 3>), grid, blk, 0, stream, (const T *)qkv, (T *)out, N, D, H, DH, scale, qblocks); break;
    case 4: hipLaunchKernelGGL((attention_generic_kernel<T, 4>), grid, blk, 0, stream, (const T *)qkv, (T *)out, N, D, H, DH, scale, qblocks); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Attention of the class token alone (last encoder layer of a classifier).  vit.cpp:910-911 keeps row 0 of the last layer's output and
// nothing else; inside a layer a token's row depends on the other tokens only through their k and v (vit.cpp:848-858).  So in the last layer
// only q of token 0 meets K and V of all tokens, and the output projection and the MLP after it run on ONE row per image (engine.cpp).
// One workgroup of four waves per (image, head).  A row of the head's K (or V) slice is NP 16-byte pieces; lane = (row group g, piece p): 64 / NP rows
// per wave and step, the waves interleaved.
//   pass 1: s_j = q . k_j in f32 (shuffle sum over the NP lanes of a row) -> LDS, and the row maximum;
//   pass 2: e_j = the kernels' numerator rule (AttnExpRt: fp16 table semantics for F16, rounded bf16 otherwise), o += e_j v_j in f32, sum of e_j;
//           the row groups are combined by shuffles, the four waves through LDS in index order; o / sum is rounded once to the operand type.
// PLANES: the F16 parity mode's two-plane q, k, v (value = hi + lo / 2048, EPI_BIAS_HILO) -- here the products are plain f32 FMAs.
// The workgroup of head h also copies the residual-stream slice X[b * N][h * DH ..] into the compact rows xc[b][..] the tail GEMMs work on.
// ------------------------------------------------------------------------------------------------
template <typename T, int NP, bool PLANES>
__global__ __launch_bounds__(256) void attention_cls_kernel(const T *__restrict__ qkv, long lo_off, T *__restrict__ out, const float *__restrict__ x, float *__restrict__ xc,
                                                            int N, int D, int H, float scale) {
    extern __shared__ float cls_sc[];                  // [N] raw scores, then [4] wave maxima, [4] wave sums, [4][DH] wave partial outputs
    typedef typename Elem<T>::v8 v8;
    constexpr int DH = NP * 8, G = 64 / NP;            // G rows per wave and step; the four waves take rows 4 G apart
    const int tid = threadIdx.x, lane = tid & 63, p = lane % NP, g = lane / NP;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int item = blockIdx.x, b = item / H, h = item - b * H;
    const size_t row_el = (size_t)3 * D;
    const T *base = qkv + (size_t)b * N * row_el + (size_t)h * DH + p * 8;         // this lane's piece of q of token 0; k at + D, v at + 2 D
    float *wmax = cls_sc + N, *wsum = wmax + 4, *wacc = wsum + 4;
    auto load8 = [&](const T *ptr, float (&f)[8]) {
        const v8 a = *(const v8 *)ptr;
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = (float)a[e];
        if (PLANES) {
            const v8 l = *(const v8 *)(ptr + lo_off);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = __builtin_fmaf((float)l[e], 1.0f / 2048.0f, f[e]);
        }
    };
    if (xc && tid < DH / 4) *(f32x4 *)(xc + (size_t)b * D + h * DH + tid * 4) = *(const f32x4 *)(x + (size_t)b * N * D + h * DH + tid * 4);
    float q[8];
    load8(base, q);
    float mx = -INFINITY;
    const int steps2 = (N + 8 * G - 1) / (8 * G);      // two steps per trip: both rows' loads are in flight together
    for (int it = 0; it < steps2; ++it) {
        const int ra = (it * 8 + wave) * G + g, rb = ra + 4 * G;
        float ka[8], kb[8];
        load8(base + D + (size_t)min(ra, N - 1) * row_el, ka);
        load8(base + D + (size_t)min(rb, N - 1) * row_el, kb);
        float sa = 0.0f, sb = 0.0f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { sa = __builtin_fmaf(q[e], ka[e], sa); sb = __builtin_fmaf(q[e], kb[e], sb); }
#pragma unroll
        for (int o = 1; o < NP; o <<= 1) { sa += __shfl_xor(sa, o); sb += __shfl_xor(sb, o); }
        if (ra < N) { mx = fmaxf(mx, sa); if (p == 0) cls_sc[ra] = sa; }
        if (rb < N) { mx = fmaxf(mx, sb); if (p == 0) cls_sc[rb] = sb; }
    }
#pragma unroll
    for (int o = NP; o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (lane == 0) wmax[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    const float kk = AttnExpRt<T>::k(scale), nmx = -kk * mx;
    float acc[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}, sum = 0.0f;
    for (int it = 0; it < steps2; ++it) {
        const int ra = (it * 8 + wave) * G + g, rb = ra + 4 * G, rac = min(ra, N - 1), rbc = min(rb, N - 1);
        float va[8], vb[8];
        load8(base + 2 * D + (size_t)rac * row_el, va);
        load8(base + 2 * D + (size_t)rbc * row_el, vb);
        const typename Pair<T>::v2 e2 = AttnExpRt<T>::pair(cls_sc[rac], cls_sc[rbc], nmx, kk);
        const float ea = ra < N ? (float)e2[0] : 0.0f, eb = rb < N ? (float)e2[1] : 0.0f;
        sum += ea; sum += eb;
#pragma unroll
        for (int i = 0; i < 8; ++i) { acc[i] = __builtin_fmaf(ea, va[i], acc[i]); acc[i] = __builtin_fmaf(eb, vb[i], acc[i]); }
    }
#pragma unroll
    for (int o = NP; o < 64; o <<= 1) {
        sum += __shfl_xor(sum, o);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += __shfl_xor(acc[i], o);
    }
    if (g == 0) {
        if (p == 0) wsum[wave] = sum;
#pragma unroll
        for (int i = 0; i < 8; ++i) wacc[wave * DH + p * 8 + i] = acc[i];
    }
    __syncthreads();
    if (tid < NP) {                                    // lane = piece: waves combined in index order
        const float inv = 1.0f / (((wsum[0] + wsum[1]) + wsum[2]) + wsum[3]);
        v8 o8;
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            const int c = tid * 8 + i;
            const float a0 = ((wacc[c] + wacc[DH + c]) + wacc[2 * DH + c]) + wacc[3 * DH + c];
            const float a1 = ((wacc[c + 1] + wacc[DH + c + 1]) + wacc[2 * DH + c + 1]) + wacc[3 * DH + c + 1];
            const typename Pair<T>::v2 pr = round_pair<T>(a0 * inv, a1 * inv);
            o8[i] = pr[0]; o8[i + 1] = pr[1];
        }
        *(v8 *)(out + (size_t)b * D + h * DH + tid * 8) = o8;
    }
}
bool attention_cls_supports(int N, int D, int H) {
    if (H <= 0 || D % H || N <= 0 || N > 15360) return false;      // scores + the waves' partial sums stay inside 64 KiB of LDS
    const int dh = D / H;
    return dh == 8 || dh == 16 || dh == 32 || dh == 64 || dh == 128;
}
template <typename T, bool PLANES>
static hipError_t launch_attention_cls_t(const void *qkv, long lo_off, void *out, const float *x, float *xc, int n_img, int N, int D, int H, hipStream_t stream) {
    const int np = D / H / 8;
    const float scale = 1.0f / sqrtf((float)(D / H));
    const dim3 grid((unsigned)((size_t)n_img * H)), blk(256);
    const size_t lds = ((size_t)N + 8 + 4 * (D / H)) * sizeof(float);
#define VITX_CLS(NP) hipLaunchKernelGGL((attention_cls_kernel<T, NP, PLANES>), grid, blk, lds, stream, (const T *)qkv, lo_off, (T *)out, x, xc, N, D, H, scale)
    switch (np) {
    case 1: VITX_CLS(1); break;
    case 2: VITX_CLS(2); break;
    case 4: VITX_CLS(4); break;
    case 8: VITX_CLS(8); break;
    case 16: VITX_CLS(16); break;
    default: return hipErrorInvalidValue;
    }
#undef VITX_CLS
    return hipGetLastError();
}
hipError_t launch_attention_cls(int dtype, const void *qkv, long lo_off, void *out, const float *x, float *xc, int n_img, int N, int D, int H, hipStream_t stream) {
    if (!attention_cls_supports(N, D, H) || n_img <= 0) return hipErrorInvalidValue;
    if (lo_off && dtype != DT_F16) return hipErrorInvalidValue;
    if (dtype == DT_F16) return lo_off ? launch_attention_cls_t<_Float16, true>(qkv, lo_off, out, x, xc, n_img, N, D, H, stream) : launch_attention_cls_t<_Float16, false>(qkv, 0, out, x, xc, n_img, N, D, H, stream);
    return launch_attention_cls_t<__bf16, false>(qkv, 0, out, x, xc, n_img, N, D, H, stream);
}

bool attention_persist_supports(int n_img, int N, int D) { return N > 192 && N <= 224 && (size_t)n_img * N * 3 * D * 2 < 0xf0000000u; }

static const int kAttnNkt[] = {1, 2, 3, 4, 5, 6, 7, 9, 19};      // instantiated key-tile counts (tokens = 32 * nkt, rounded up)
bool attention_single_pass_supports(int N) {
    const int nkt = (N + 31) / 32;
    for (int k : kAttnNkt) if (k == nkt) return true;
    return false;
}
bool attention_supports(int N, int D, int H) { return N > 0 && (D == H * 64 || attention_generic_supports(D, H)); }      // any token count; head_dim 64 (tuned kernels) or any multiple of 8 up to 128
// Kernel choice (measured, 128 x 12 heads bf16: 197 tokens 52 vs 55 us, 257 tokens 77 vs 92 us single-pass vs pipelined;
// 64 x 16 heads x 577 tokens 282 vs 241 us -- profiles/r02_attention.txt):
//   193..224 tokens: the persistent single-pass kernel at EVERY batch size (its 16x16x32 products group the f32 sums differently from the
//   32x32x16 kernels: one kernel per token count keeps an image's result independent of the batch it arrives in); otherwise single-pass
//   (all scores in registers) up to 288 tokens where instantiated, the pipelined two-pass kernel for everything else.
// t.attn_kernel (vitx_op_attention_ex, tests): ATTN_SINGLE / ATTN_FLOW / ATTN_PERSIST force one family.
hipError_t launch_attention(const Tuning &t, int dtype, const void *qkv, void *out, int n_img, int N, int D, int H, hipStream_t stream) {
    if (!attention_supports(N, D, H)) return hipErrorInvalidValue;
    if (D != H * 64) return dtype == DT_F16 ? launch_attention_generic<_Float16>(qkv, out, n_img, N, D, H, stream) : launch_attention_generic<__bf16>(qkv, out, n_img, N, D, H, stream);
    // 193..224 tokens: the persistent single-pass kernel (K/V of the next item by LDS-DMA under the current item's softmax)
    if ((t.attn_kernel == ATTN_PERSIST || t.attn_kernel == ATTN_AUTO) && N > 192 && N <= 224) {
        // 32-bit buffer offsets bound one launch (~4.4 k ViT-B images): a larger sub-batch is cut into several launches of the SAME kernel rather than
        // handed to another family (whose f32 sums are grouped differently: an image's result must not depend on the batch it arrives in -- r03 advisor)
        const size_t per_img = (size_t)N * 3 * D * 2;
        const int max_img = (int)std::min<size_t>((size_t)n_img, (0xf0000000u - 1) / per_img);
        if (max_img < 1) return hipErrorInvalidValue;
        for (int i0 = 0; i0 < n_img; i0 += max_img) {
            const int ni = std::min(max_img, n_img - i0);
            const char *q = (const char *)qkv + (size_t)i0 * per_img; char *o = (char *)out + (size_t)i0 * N * D * 2;
            const hipError_t e = dtype == DT_F16 ? launch_attention_persist<_Float16>(q, o, ni, N, D, H, t.attn_grid > 0 ? t.attn_grid : t.n_cu, stream, t.attn_flags)
                                                 : launch_attention_persist<__bf16>(q, o, ni, N, D, H, t.attn_grid > 0 ? t.attn_grid : t.n_cu, stream, t.attn_flags);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }
    if (t.attn_kernel == ATTN_PERSIST) return hipErrorInvalidValue;
    if (t.attn_kernel == ATTN_STREAM) return launch_attention_stream(dtype, false, qkv, out, n_img, N, D, H, 0, stream);
    const bool single = attention_single_pass_supports(N) && (N <= 288 || t.attn_kernel == ATTN_SINGLE);
    if (t.attn_kernel == ATTN_FLOW || !single)
        return dtype == DT_F16 ? launch_attention_flow<_Float16>(qkv, out, n_img, N, D, H, stream, t.attn_flags) : launch_attention_flow<__bf16>(qkv, out, n_img, N, D, H, stream, t.attn_flags);
    return dtype == DT_F16 ? launch_attention_t<_Float16>(qkv, out, n_img, N, D, H, stream) : launch_attention_t<__bf16>(qkv, out, n_img, N, D, H, stream);
}

// ------------------------------------------------------------------------------------------------
// Per-device launch state (see Tuning in kernels.h).
// ------------------------------------------------------------------------------------------------
static hipError_t prepare_device_kernels(const Tuning &t) {
    GemmArgs none{};
    hipError_t e;
    for (int dt = 0; dt < 2; ++dt) {
        for (int epi = 0; epi <= EPI_BIAS_HILO; ++epi) {
            for (int cfg : {945, 445, 245, 122}) if ((e = launch_gemm_ring(t, dt, epi, none, cfg, nullptr, true)) != hipSuccess) return e;
            if ((e = launch_gemm_pp(dt, epi, none, t.n_cu, nullptr, 0, true)) != hipSuccess) return e;
#ifdef VITX_LAB
            for (int nw : {4, 8}) if ((epi == EPI_BIAS || epi == EPI_BIAS_GELU) && (e = launch_gemm_w4(dt, epi, none, t.n_cu, nullptr, 0, true, nw)) != hipSuccess) return e;
#endif
            if ((e = (dt == DT_F16 ? launch_gemm_t<_Float16, true>(epi, none, nullptr, true) : launch_gemm_t<__bf16, true>(epi, none, nullptr, true))) != hipSuccess) return e;
        }
        if (dt == 0 && (e = launch_patch_embed(0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, 0, 0, nullptr, true)) != hipSuccess) return e;
        if ((e = (dt == DT_F16 ? launch_attention_flow<_Float16>(nullptr, nullptr, 0, 64, 64, 1, nullptr) : launch_attention_flow<__bf16>(nullptr, nullptr, 0, 64, 64, 1, nullptr))) != hipSuccess) return e;
        if ((e = launch_attention_stream(dt, false, nullptr, nullptr, 0, 64, 64, 1, 0, nullptr)) != hipSuccess) return e;
        if (dt == DT_F16 && (e = launch_attention_stream(dt, true, nullptr, nullptr, 0, 64, 64, 1, 0, nullptr)) != hipSuccess) return e;
        if ((e = (dt == DT_F16 ? launch_attention_persist<_Float16>(nullptr, nullptr, 0, 224, 64, 1, t.n_cu, nullptr) : launch_attention_persist<__bf16>(nullptr, nullptr, 0, 224, 64, 1, t.n_cu, nullptr))) != hipSuccess) return e;
        for (int nkt : kAttnNkt) {
            e = dt == DT_F16 ? launch_attention_t<_Float16>(nullptr, nullptr, 0, nkt * 32, 64, 1, nullptr) : launch_attention_t<__bf16>(nullptr, nullptr, 0, nkt * 32, 64, 1, nullptr);
            if (e != hipSuccess) return e;
        }
    }
    return hipSuccess;
}

const Tuning *tuning_for_device(int device) {
    static std::mutex mu;
    static std::vector<std::unique_ptr<Tuning>> table;      // one entry per device, never moved or freed
    if (device < 0 && hipGetDevice(&device) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if ((size_t)device < table.size() && table[device]) return table[device].get();
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) return nullptr;
    if (cur != device && hipSetDevice(device) != hipSuccess) return nullptr;
    std::unique_ptr<Tuning> t(new Tuning());
    t->device = device;
    if (hipDeviceGetAttribute(&t->n_cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || t->n_cu <= 0) t->n_cu = 256;
    if (hipDeviceGetAttribute(&t->n_xcd, hipDeviceAttributeNumberOfXccs, device) != hipSuccess || t->n_xcd <= 0) { (void)hipGetLastError(); t->n_xcd = 0; }      // unknown: no LayerNorm fusion
#ifdef VITX_LAB      // the laboratory build (tools/) reads its experiment switches from the environment; the product library reads none of them
    auto env_int = [](const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; };
    if (const char *e = getenv("VITX_GEMM_CFG")) t->gemm_cfg = !strcmp(e, "pp") ? 1 : atoi(e);
    t->gemm_split = env_int("VITX_GEMM_SPLIT", 0);
    t->gemm_balance = env_int("VITX_GEMM_BALANCE", 1);
    t->group_m = env_int("VITX_GROUP_M", 0);
    t->skinny_tiles = env_int("VITX_SKINNY_TILES", 128);
    t->pp_flags = env_int("VITX_PP_FLAGS", 0);
    t->gemm_dbg = env_int("VITX_GEMM_DBG", 0);
    t->attn_kernel = env_int("VITX_ATTN_KERNEL", 0);
    t->attn_grid = env_int("VITX_ATTN_GRID", 0);
#endif
    const hipError_t e = prepare_device_kernels(*t);
    if (cur != device) (void)hipSetDevice(cur);
    if (e != hipSuccess) return nullptr;
    if ((size_t)device >= table.size()) table.resize(device + 1);
    table[device] = std::move(t);
    return table[device].get();
}

// ------------------------------------------------------------------------------------------------
// Class softmax (ggml_soft_max, vit.cpp:931): max, e_i = round(expf(round(x_i - max))), p = e * (1/sum).
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void softmax_kernel(const float *__restrict__ logits, float *__restrict__ probs, int cols, int ld) {
    __shared__ float red[4];
    const float *x = logits + (size_t)blockIdx.x * ld;
    float *p = probs + (size_t)blockIdx.x * cols;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float mx = -INFINITY;
    for (int i = tid; i < cols; i += 256) mx = fmaxf(mx, x[i]);
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (lane == 0) red[wv] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.0f;
    for (int i = tid; i < cols; i += 256) { const float e = rnd<T>(expf(rnd<T>(x[i] - mx))); p[i] = e; sum += e; }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane == 0) red[wv] = sum;
    __syncthreads();
    const float inv = 1.0f / ((red[0] + red[1]) + (red[2] + red[3]));
    for (int i = tid; i < cols; i += 256) p[i] *= inv;
}
hipError_t launch_softmax(int dtype, const float *logits, float *probs, int rows, int cols, int ld, hipStream_t stream) {
    if (dtype == DT_F16) hipLaunchKernelGGL(softmax_kernel<_Float16>, dim3(rows), dim3(256), 0, stream, logits, probs, cols, ld);
    else hipLaunchKernelGGL(softmax_kernel<__bf16>, dim3(rows), dim3(256), 0, stream, logits, probs, cols, ld);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Top-k of every probability row (vit_predict's sort, vit.cpp:1043-1057): one wave per row, k selection passes; pass i takes the
// largest entry that comes after pass i - 1's in the order (probability descending, class index ascending) -- no scratch, no ties lost.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void topk_kernel(const float *__restrict__ probs, int rows, int cols, int k, float *__restrict__ out) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *p = probs + (size_t)row * cols;
    float pv = INFINITY; int pi = -1;
    for (int it = 0; it < k; ++it) {
        float bv = -INFINITY; int bi = 0x7fffffff;
        for (int i = lane; i < cols; i += 64) {
            const float v = p[i];
            const bool after = v < pv || (v == pv && i > pi);            // not yet taken
            if (after && (v > bv || (v == bv && i < bi))) { bv = v; bi = i; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o); const int oi = __shfl_xor(bi, o);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { out[((size_t)row * k + it) * 2] = bv; ((int *)out)[((size_t)row * k + it) * 2 + 1] = bi; }
        pv = bv; pi = bi;
    }
}
__global__ void spin_kernel(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}
hipError_t launch_spin(int microseconds, hipStream_t stream) {
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, stream, (long long)microseconds * 100);
    return hipGetLastError();
}
// the same, leaving its own first and last reading of the 100 MHz wall clock in stamps[0..1]: a kernel whose duration the device itself
// states, the yardstick for what a HIP-event bracket adds to a launch (vitx_profile_bracket_us)
__global__ void spin_stamp_kernel(long long ticks, long long *stamps) {
    const long long t0 = wall_clock64();
    long long t1 = t0;
    while (t1 - t0 < ticks) { __builtin_amdgcn_s_sleep(2); t1 = wall_clock64(); }
    if (threadIdx.x == 0) { stamps[0] = t0; stamps[1] = t1; }
}
hipError_t launch_spin_stamp(int microseconds, long long *stamps, hipStream_t stream) {
    hipLaunchKernelGGL(spin_stamp_kernel, dim3(1), dim3(64), 0, stream, (long long)microseconds * 100, stamps);
    return hipGetLastError();
}
hipError_t launch_topk(const float *probs, int rows, int cols, int k, void *out_pairs, hipStream_t stream) {
    if (rows <= 0 || cols <= 0 || k <= 0 || k > cols) return hipErrorInvalidValue;
    hipLaunchKernelGGL(topk_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, probs, rows, cols, k, (float *)out_pairs);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Device-side vit_image_preprocess (vit.cpp:289-305; bicubic 204-287, bilinear 130-196): u8 HWC
// [n][ny][nx][3] -> f32 HWC [n][S][S][3], one thread per output pixel.  Operation for operation the
// host version in preprocess.cpp (double cubic coefficients narrowed to float, float polynomial,
// roundf / clamp / narrow to u8, (q - mean) / std with IEEE division; the library is built with
// -ffp-contract=off), so the two agree bit for bit.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float pp_norm(float v, int k) {
    const float mean = k == 0 ? 123.675f : (k == 1 ? 116.280f : 103.530f);
    const float sd = k == 0 ? 58.395f : (k == 1 ? 57.120f : 57.375f);
    const unsigned char q = (unsigned char)fminf(fmaxf(roundf(v), 0.0f), 255.0f);
    return ((float)q - mean) / sd;
}
__device__ __forceinline__ float pp_cubic(float p0, float p1, float p2, float p3, float t) {
    const float d0 = p0 - p1, d2 = p2 - p1, d3 = p3 - p1;
    const float a1 = (float)(-1.0 / 3 * d0 + d2 - 1.0 / 6 * d3);
    const float a2 = (float)(1.0 / 2 * d0 + 1.0 / 2 * d2);
    const float a3 = (float)(-1.0 / 6 * d0 - 1.0 / 2 * d2 + 1.0 / 6 * d3);
    return p1 + a1 * t + a2 * t * t + a3 * t * t * t;
}
__device__ __forceinline__ int pp_clamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

template <bool BICUBIC>
__global__ __launch_bounds__(256) void preprocess_kernel(const unsigned char *__restrict__ src, float *__restrict__ dst, int n, int nx, int ny, int S) {
    const long total = (long)n * S * S;
    for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long)gridDim.x * blockDim.x) {
        const int b = (int)(id / ((long)S * S)), rem = (int)(id - (long)b * S * S), i = rem / S, j = rem - i * S;
        const unsigned char *im = src + (size_t)b * nx * ny * 3;
        float *o = dst + (size_t)id * 3;
        if (BICUBIC) {
            const float tx = (float)nx / (float)S, ty = (float)ny / (float)S;
            const int y = (int)(ty * i), x = (int)(tx * j);
            const float dy = ty * i - y, dx = tx * j - x;
            const int x0 = pp_clamp(x - 1, 0, nx - 1), x1 = pp_clamp(x, 0, nx - 1), x2 = pp_clamp(x + 1, 0, nx - 1), x3 = pp_clamp(x + 2, 0, nx - 1);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float C[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const unsigned char *r = im + (size_t)pp_clamp(y - 1 + jj, 0, ny - 1) * nx * 3 + k;
                    C[jj] = pp_cubic(r[x0 * 3], r[x1 * 3], r[x2 * 3], r[x3 * 3], dx);
                }
                o[k] = pp_norm(pp_cubic(C[0], C[1], C[2], C[3], dy), k);
            }
        } else {
            const float xs = nx / (float)S, ys = ny / (float)S;
            const float sy = (i + 0.5f) * ys - 0.5f, sx = (j + 0.5f) * xs - 0.5f;
            const int y0 = sy < 0.0f ? 0 : (int)floorf(sy), y1 = y0 + 1 < ny - 1 ? y0 + 1 : ny - 1;
            const int x0 = sx < 0.0f ? 0 : (int)floorf(sx), x1 = x0 + 1 < nx - 1 ? x0 + 1 : nx - 1;
            const float dy = sy - y0, dx = sx - x0;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v00 = im[3 * ((size_t)y0 * nx + x0) + c], v01 = im[3 * ((size_t)y0 * nx + x1) + c];
                const float v10 = im[3 * ((size_t)y1 * nx + x0) + c], v11 = im[3 * ((size_t)y1 * nx + x1) + c];
                const float v0 = v00 * (1.0f - dx) + v01 * dx, v1 = v10 * (1.0f - dx) + v11 * dx;
                o[c] = pp_norm(v0 * (1.0f - dy) + v1 * dy, c);
            }
        }
    }
}
hipError_t launch_preprocess(const void *u8, float *out, int n, int nx, int ny, int S, int bicubic, hipStream_t stream) {
    const long total = (long)n * S * S;
    const int blocks = (int)std::min<long>((total + 255) / 256, 256L * 64);
    if (bicubic) hipLaunchKernelGGL(preprocess_kernel<true>, dim3(blocks), dim3(256), 0, stream, (const unsigned char *)u8, out, n, nx, ny, S);
    else hipLaunchKernelGGL(preprocess_kernel<false>, dim3(blocks), dim3(256), 0, stream, (const unsigned char *)u8, out, n, nx, ny, S);
    return hipGetLastError();
}

}  // namespace vitx

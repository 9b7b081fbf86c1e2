// gemm_pp.hip -- "ping-pong" persistent MFMA GEMM for gfx950: the wide-tile kernel of the ViT forward path.
//
//   C[M][N] = A[M][K] . W[N][K]^T  (+ fused epilogue), A/W fp16 or bf16, f32 accumulate
//   (ggml_mul_mat at /root/reference/vit.cpp:772,820,868,889,896 with the bias / GELU / residual / pos-embed ops fused).
//
// Structure (DESIGN.md "GEMM"):
//   * 512 threads = 8 waves as 2(M) x 4(N); tile 256x256, BK = 64; each wave owns 128x64 of C as 8x4 accumulators of
//     v_mfma_f32_16x16x32 (r02f; the 4x2 v_mfma_f32_32x32x16 form of r02a-e is FLAGS 65536: same cycles, 11 % more energy per
//     flop).  One persistent workgroup per CU walks its tiles and keeps ONE operand stream running across tile boundaries.
//   * The two wave rows (waves 0-3 / 4-7: one wave of each per SIMD) run ONE BARRIER APART ("ping-pong"): while one
//     group issues the MFMAs of a phase (16 of 16x16x32), the other issues its LDS fragment reads and LDS-DMA for its own phase, so
//     every SIMD always has one wave in the matrix pipe and one in the memory pipes.  s_setprio(1) brackets the MFMAs.
//   * A K-tile is 4 phases, one C quadrant (64x32 per wave, K = 64 -> 256 MFMA cycles) each, in the snake order
//     C00, C01, C11, C10 so every operand fragment is read from LDS exactly once: 12 / 4 / 8 / 0 ds_read_b128.
//   * LDS = 2 buffers x [A0 | A1 | B0 | B1] half-tiles of 16 KiB (128 rows x 128 B).  "A0" holds, for both wave rows,
//     the first 64 of the wave's 128 rows (B0: for the four wave columns, the first 32 of the wave's 64 columns), so a
//     half-tile is read in exactly one phase and can be re-staged two phases later.  One half-tile is staged per
//     phase by LDS-DMA (global_load_lds dwordx4, 2 per thread), 5 phases ahead of its first read; a counted
//     s_waitcnt vmcnt(8) per phase leaves the four youngest stages in flight across the raw s_barriers.
//   * 128-B LDS rows, two rows per 256-B bank line, 16-B slots XOR-ed with (line & 15): conflict-free ds_read_b128;
//     the DMA image is lane-linear, so the permutation is applied to the per-lane global SOURCE address.
//   * Products are "swapped" (mfma(W fragment, A fragment)): each lane ends up with ONE row of C and 4 consecutive
//     columns per accumulator group, so the epilogue moves 8/16 contiguous bytes per lane per store instead of 2.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "device_common.h"
#include "kernels.h"
#include "epilogue16.h"

// Register cap of the kernel (hipcc doubles amdgpu_num_vgpr on gfx90a+: arch + accumulator halves, so 128 = all 256 registers of a
// 2-waves-per-SIMD kernel).  r02 experiment: 116 (= 232) leaves 48 VGPRs per SIMD free, exactly one LayerNorm wave, so the other
// sub-batch stream's LayerNorm could co-reside with a persistent GEMM -- measured 2.7 % SLOWER on the whole forward
// (10.91 vs 10.61 ms/step, interleaved A/B): the spills it forces cost more than the overlap buys.
#ifndef PP_MAX_VGPR
#define PP_MAX_VGPR 128
#endif

namespace vitx {

namespace pp {
constexpr int BM = 256, BN = 256, BK = 64;
constexpr int HALF = 16384;             // one half-tile image: 128 rows x 128 B
constexpr int LDS = 8 * HALF;           // 128 KiB operand ring: [A0 A1 of buffer 0 | A0 A1 of buffer 1 | B0 B1 of buffer 0 | B0 B1 of buffer 1]
// all four A half-tiles lie in the first 64 KiB and all four B half-tiles in the second, so every fragment read is
// "one per-lane base register + a 16-bit immediate": no per-buffer address copies (8 VGPRs and 8 adds per K-tile less)
__host__ __device__ constexpr int off_a(int buf, int h) { return (buf * 2 + h) * HALF; }
__host__ __device__ constexpr int off_b(int buf, int h) { return (4 + buf * 2 + h) * HALF; }
constexpr int LDS_ALL = LDS + 8 * 4096; // + one 4 KiB epilogue patch per wave = all 160 KiB
constexpr int GROUP_M = 8;
constexpr int STAGE_OPS = 2;            // LDS-DMA instructions per thread per half-tile
constexpr int LEAD = 4;                 // stages allowed in flight past a phase's wait
}  // namespace pp

template <int N> __device__ __forceinline__ void pp_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// counted vector-memory wait + every LDS read of this wave landed (the two-burst schedule re-stages a half-tile one burst after
// its last read: the reads must be complete BEFORE the barrier that releases the other wave row's stage issue)
template <int N> __device__ __forceinline__ void pp_wait_vm_lgkm() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void pp_barrier() { asm volatile("s_barrier" ::: "memory"); }

// ---- epilogue of one wave's 128 x 64 block.  Swapped-product C layout: lane -> row l31 of each 32-row block,
// register r -> column (r&3) + 8*(r>>2) + 4*(lane>>5) of each 32-column block.
template <typename T, int EPI, bool FULL>
__device__ __forceinline__ void pp_epilogue(const GemmArgs &g, f32x16 (&acc)[4][2], int row0 /* + i*32 */, int ncol /* wave-uniform first column */, int hh) {
    typedef typename Elem<T>::v4 v4;
    typedef const __attribute__((address_space(4))) float *cptr;     // constant address space: wave-uniform -> s_load (no vmcnt traffic)
    const int col0 = ncol + 4 * hh;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            // bias of the 8 columns of this accumulator group: lanes 0-31 take the first four, lanes 32-63 the last four
            // (the bias buffer is padded to the N tile, so reads beyond N stay inside it)
            cptr cb = (cptr)(g.bias + ncol + j * 32 + rg * 8);
            f32x4 bv;
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float lo = cb[e], hi = cb[4 + e]; bv[e] = hh ? hi : lo; }
            const int c = col0 + j * 32 + rg * 8;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = row0 + i * 32;
                const bool ok = FULL || (row < g.M_real && c < g.N);         // N % 4 == 0: a 4-column group is all in or all out
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][rg * 4 + e] + bv[e];
                if constexpr (EPI == EPI_BIAS) {
                    if (ok) *(v4 *)((T *)g.out + (size_t)row * g.ldo + c) = __builtin_convertvector(v, v4);
                } else if constexpr (EPI == EPI_BIAS_GELU) {
                    // bias, round to the operand type (ggml's fp16 LUT input), tanh-GELU, round (LUT output)
                    const typename Pair<T>::v2 x01 = round_pair<T>(v[0], v[1]), x23 = round_pair<T>(v[2], v[3]);
                    const f32x2 y01 = gelu_tanh2(f32x2{(float)x01[0], (float)x01[1]}), y23 = gelu_tanh2(f32x2{(float)x23[0], (float)x23[1]});
                    const typename Pair<T>::v2 o01 = round_pair<T>(y01[0], y01[1]), o23 = round_pair<T>(y23[0], y23[1]);
                    if (ok) *(v4 *)((T *)g.out + (size_t)row * g.ldo + c) = v4{o01[0], o01[1], o23[0], o23[1]};
                } else if constexpr (EPI == EPI_BIAS_RESID) {
                    if (ok) { f32x4 *p = (f32x4 *)((float *)g.out + (size_t)row * g.ldo + c); *p = v + *p; }
                } else if constexpr (EPI == EPI_BIAS_F32) {
                    if (ok) *(f32x4 *)((float *)g.out + (size_t)row * g.ldo + c) = v;
                } else {   // EPI_PATCH: patch row -> token row (+1 per image for the cls slot), + pos_embed
                    if (ok) {
                        const int b = row / g.tpi, t = row - b * g.tpi;
                        const f32x4 pe = *(const f32x4 *)(g.pos + (size_t)(t + 1) * g.ldo + c);
                        *(f32x4 *)((float *)g.out + ((size_t)row + b + 1) * g.ldo + c) = v + pe;
                    }
                }
            }
        }
    }
}

// ---- full-tile epilogue: EXACT number of vector-memory instructions (pp_epi_stores) and whole-row stores.
//   * every store is one buffer_store_dwordx4 issued unconditionally, so the K loop of the next tile can skip over them
//     with a counted vmcnt instead of draining them;
//   * the accumulators hold one ROW per lane (a store straight from them touches 64 different 128-byte lines per
//     instruction and the texture-address path serialises on lines: ~9k cycles per tile, r02 lab), so each 32-row block
//     goes through a wave-private 4 KiB LDS patch (the 32 KiB the operand ring leaves free): written in the MFMA layout,
//     read back with 8 lanes per 128-byte row, stored as whole lines (8 lines per instruction).
//     Patch rows are 128 B; 16-byte slots are XOR-ed with (row & 7): conflict-free reads, <= 2-way writes.
template <int EPI> __host__ __device__ constexpr int pp_epi_stores() { return (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) ? 16 : 32; }

template <typename T, int EPI, bool NOSTORE = false, int AUX = 0>
__device__ __forceinline__ void pp_epilogue_full(f32x16 (&acc)[4][2], __amdgpu_buffer_rsrc_t ro, char *patch /* wave-private 4 KiB; its first 256 B hold the bias of the wave's 64 columns */,
                                                  int voff /* this lane's byte offset in row layout */, int soff /* tile origin */, int soff8 /* 8 rows */, int lane) {
    const int l31 = lane & 31, hh = lane >> 5;
    const int wr_row = l31 * 128, x16 = (l31 & 7) * 16;                                     // MFMA layout: this lane's patch row
    const int rd_off = (lane >> 3) * 128 + (((lane & 7) ^ ((lane >> 3) & 7)) * 16);         // row layout: row (lane>>3) + 8t, 16-byte piece lane&7
    // bias of this lane's 32 columns (8 groups q of 4: columns q*8 + 4hh ..), staged into the patch by LDS-DMA during the K loop
    f32x4 bq[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) bq[q] = *(const f32x4 *)(patch + q * 32 + hh * 16);
    pp_lds_fence();
    if constexpr (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) {
        typedef typename Pair<T>::v2 v2;
        // one 32-row block: bias, (GELU,) pack -> 16 dwords per lane; the NEXT block is computed while this one's LDS writes land
        auto compute = [&](int i, u32x2 (&w)[8]) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {          // q = j*4 + rg: columns q*8 + 4hh .. +3 -> 8 bytes of slot q
                const int j = q >> 2, rg = q & 3;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][rg * 4 + e] + bq[q][e];
                v2 p0 = round_pair<T>(v[0], v[1]), p1 = round_pair<T>(v[2], v[3]);
                if constexpr (EPI == EPI_BIAS_GELU) {      // round to the operand type (ggml's fp16 LUT input), tanh-GELU, round (LUT output)
                    const f32x2 y0 = gelu_tanh2(f32x2{(float)p0[0], (float)p0[1]}), y1 = gelu_tanh2(f32x2{(float)p1[0], (float)p1[1]});
                    p0 = round_pair<T>(y0[0], y0[1]); p1 = round_pair<T>(y1[0], y1[1]);
                }
                w[q] = u32x2{__builtin_bit_cast(unsigned, p0), __builtin_bit_cast(unsigned, p1)};
            }
        };
        u32x2 w[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            compute(i, w);
#pragma unroll
            for (int q = 0; q < 8; ++q) *(u32x2 *)(patch + wr_row + ((q * 16) ^ x16) + hh * 8) = w[q];
            pp_lds_fence();
            u32x4 d[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) d[t] = *(const u32x4 *)(patch + t * 1024 + rd_off);
            pp_lds_fence();
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if constexpr (NOSTORE) asm volatile("" :: "v"(d[t])); else
                pp_store_b128(d[t], ro, voff, soff + (i * 4 + t) * soff8);
            }
        }
    } else {       // f32 outputs: one 32 x 32 accumulator block (4 KiB) per pass
        u32x4 res[2][4];                            // residual rows of the current and the next pass (loads run one pass ahead)
        auto load_res = [&](int c, u32x4 (&dst)[4]) {
            const int i = c >> 1, j = c & 1;
#pragma unroll
            for (int t = 0; t < 4; ++t) dst[t] = __builtin_amdgcn_raw_buffer_load_b128(ro, voff + j * 128, soff + (i * 4 + t) * soff8, 0);
        };
        if constexpr (EPI == EPI_BIAS_RESID) load_res(0, res[0]);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int i = c >> 1, j = c & 1;
            if constexpr (EPI == EPI_BIAS_RESID) { if (c + 1 < 8) load_res(c + 1, res[(c + 1) & 1]); }
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][rg * 4 + e] + bq[j * 4 + rg][e];
                *(f32x4 *)(patch + wr_row + (((rg * 2 + hh) * 16) ^ x16)) = v;
            }
            pp_lds_fence();
            f32x4 d[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) d[t] = *(const f32x4 *)(patch + t * 1024 + rd_off);
            pp_lds_fence();
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if constexpr (EPI == EPI_BIAS_RESID) d[t] = d[t] + __builtin_bit_cast(f32x4, res[c & 1][t]);     // (acc + bias) + x, the reference's order (vit.cpp:868-873)
                pp_store_b128<AUX>(__builtin_bit_cast(u32x4, d[t]), ro, voff + j * 128, soff + (i * 4 + t) * soff8);
            }
        }
    }
}

// LayerNorm of rows m0 .. m0 + nrows - 1 of the f32 matrix the GEMM just completed (g.out, row length g.N = 256 * NV floats), written
// as operand-type rows to g.ln_out: the arithmetic of layernorm_kernel (kernels.hip) statement for statement -- per lane 4 * NV values
// (float4 i at element (i * 64 + lane) * 4), sum over i then j, xor-shuffle tree, mean, centred sum of squares, same tree,
// 1 / sqrt(var + eps), ((x - mean) * scale) * w + b, one rounding -- so the fused path is bit-identical to the stand-alone kernel
// (/root/reference/vit.cpp:808-812, 881-885).  One wave per row, four rows in flight per wave.
template <typename T>
__device__ __forceinline__ void pp_layernorm_rows(const GemmArgs &g, int m0, int nrows, int wave, int lane) {
    const int D = g.N, NV = D >> 8;
    const float *X = (const float *)g.out;
    T *U = (T *)g.ln_out;
    for (int r0 = wave * 4; r0 < nrows; r0 += 32) {
        float v[4][6][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = min(r0 + q, nrows - 1);                  // rows past the end re-read the last row and are not stored
            const float *xr = X + (size_t)(m0 + r) * g.ldo;
#pragma unroll
            for (int i = 0; i < 6; ++i) if (i < NV) { const f32x4 t = *(const f32x4 *)(xr + (i * 64 + lane) * 4); v[q][i][0] = t[0]; v[q][i][1] = t[1]; v[q][i][2] = t[2]; v[q][i][3] = t[3]; }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float sum = 0.0f;
#pragma unroll
            for (int i = 0; i < 6; ++i) if (i < NV) {
#pragma unroll
                for (int j = 0; j < 4; ++j) sum += v[q][i][j];
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
            const float mean = sum / (float)D;
            float sum2 = 0.0f;
#pragma unroll
            for (int i = 0; i < 6; ++i) if (i < NV) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[q][i][j] -= mean; sum2 += v[q][i][j] * v[q][i][j]; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) sum2 += __shfl_xor(sum2, o);
            const float scale = 1.0f / sqrtf(sum2 / (float)D + g.ln_eps);
            if (r0 + q < nrows) {
                T *yr = U + (size_t)(m0 + r0 + q) * D;
#pragma unroll
                for (int i = 0; i < 6; ++i) if (i < NV) {
                    const int idx = (i * 64 + lane) * 4;
                    typename Elem<T>::v4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { float t = v[q][i][j] * scale; t = t * g.ln_w[idx + j]; o[j] = (T)(t + g.ln_b[idx + j]); }
                    *(typename Elem<T>::v4 *)(yr + idx) = o;
                }
            }
        }
    }
}

// FLAGS (experiments, tools/gemm_lab): 1 = no s_setprio around the MFMAs, 2 = both wave rows in lock-step (no ping-pong),
// 2048 = no epilogue at all, 512 = epilogue stores drained (no counted skip), 4 = no LDS-DMA in the loop, 8 = no fragment reads, 16 = no MFMAs, 32 = s_memtime stamp after every barrier of K-tiles 4..7 of
// the first tile (written to g.pos as [block][wave][64] u32)
template <typename T, int EPI, int FLAGS>
__global__ __launch_bounds__(512, 2) __attribute__((amdgpu_num_vgpr(PP_MAX_VGPR))) void gemm_pp_kernel(GemmArgs g) {
    using namespace pp;
    typedef typename Elem<T>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;            // wave row (= ping-pong group) / wave column

    // ---- tile walk: virtual id v keeps v % 8 == bid % 8 (same XCD), then the XCD-contiguous GROUP_M raster
    const int ntm = g.M / BM, ntn = g.N_pad / BN, ntiles = ntm * ntn;
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int my_tiles = (ntiles - bid + nblk - 1) / nblk;
    const int q8 = ntiles >> 3, r8 = ntiles & 7, xcd = bid & 7;
    const int lid_base = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8);
    const int group_m = g.group_m > 0 ? g.group_m : GROUP_M;
    auto tile_origin = [&](int round, int &m0, int &n0) {
        const int v = bid + round * nblk;
        const int lid = lid_base + (v >> 3);
        const int per_group = group_m * ntn;
        const int grp = lid / per_group, within = lid - grp * per_group;
        const int gm = min(group_m, ntm - grp * group_m);
        const int tn = within / gm;
        m0 = (grp * group_m + (within - tn * gm)) * BM; n0 = tn * BN;
    };
    if (my_tiles <= 0) return;

    // ---- LDS-DMA: physical 16-B piece p = i*512 + tid of a half-tile image <-> logical (image row, slot)
    const T *A = (const T *)g.A, *W = (const T *)g.W;
    int aoff[STAGE_OPS], woff[STAGE_OPS];
#pragma unroll
    for (int i = 0; i < STAGE_OPS; ++i) {
        int rr, sl; swz_inv(i * 512 + tid, rr, sl);
        aoff[i] = ((rr >> 6) * 128 + (rr & 63)) * g.lda + sl * 8;      // image row -> tile row of half 0 (half 1: + 64 rows)
        woff[i] = ((rr >> 5) * 64 + (rr & 31)) * g.ldw + sl * 8;       // image row -> tile column of half 0 (half 1: + 32 columns)
    }
    const int a_half = 64 * g.lda, w_half = 32 * g.ldw;
    const int nkt = g.K / BK;                       // K-tiles per tile (even)

    // issue side: K-tile `is_kt` of tile round `is_round`; half-tiles go out in the order A0, B0, B1, A1
    int is_round = 0, is_kt = 0, is_a = 0, is_w = 0;
    { int m0, n0; tile_origin(0, m0, n0); is_a = m0 * g.lda; is_w = n0 * g.ldw; }
    // LDS-DMA through buffer_load ... lds: SRD + 32-bit per-lane byte offset + SGPR offset, so a stage costs two SALU adds and no VALU
    // (global_load_lds with 64-bit per-lane addresses, FLAGS 256, made the stage issue 2.5x slower: profiles/r02_gemm_pp_lab.txt)
    __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc((void *)A, 0, (int)0xffffffffu, 0x00020000);
    __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc((void *)W, 0, (int)0xffffffffu, 0x00020000);
    __amdgpu_buffer_rsrc_t rsrcO = __builtin_amdgcn_make_buffer_rsrc(g.out, 0, (int)0xffffffffu, 0x00020000);
    auto stage_a = [&](int h, int lds_off) {
        char *base = smem + lds_off + wave * 1024;
        if constexpr ((FLAGS & 256) == 0) {
            const int so = (is_a + is_kt * BK + h * a_half) * 2;
#pragma unroll
            for (int i = 0; i < STAGE_OPS; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, LPTR(base + i * 8192), 16, aoff[i] * 2, so, 0, 0);
        } else {
            const T *src = A + is_a + is_kt * BK + h * a_half;
#pragma unroll
            for (int i = 0; i < STAGE_OPS; ++i) __builtin_amdgcn_global_load_lds(GPTR(src + aoff[i]), LPTR(base + i * 8192), 16, 0, 0);
        }
    };
    auto stage_w = [&](int h, int lds_off) {
        char *base = smem + lds_off + wave * 1024;
        if constexpr ((FLAGS & 256) == 0) {
            const int so = (is_w + is_kt * BK + h * w_half) * 2;
#pragma unroll
            for (int i = 0; i < STAGE_OPS; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, LPTR(base + i * 8192), 16, woff[i] * 2, so, 0, 0);
        } else {
            const T *src = W + is_w + is_kt * BK + h * w_half;
#pragma unroll
            for (int i = 0; i < STAGE_OPS; ++i) __builtin_amdgcn_global_load_lds(GPTR(src + woff[i]), LPTR(base + i * 8192), 16, 0, 0);
        }
    };
    // after the A1 stage of a K-tile.  Past the end of the workgroup's stream the issue side stays on the last K-tile:
    // those stages re-read valid memory into LDS regions that are never read again, which keeps ONE branch-free
    // K-tile body with uniform wait counts (a peeled tail copy made hipcc spill ~230 registers).
    const int last_round = my_tiles - 1;
    auto advance = [&]() {
        if (is_kt + 1 < nkt) ++is_kt;
        else if (is_round < last_round) {
            is_kt = 0; ++is_round;
            int m0, n0; tile_origin(is_round, m0, n0); is_a = m0 * g.lda; is_w = n0 * g.ldw;
        }
    };

    // ---- fragment read addresses (bytes within a buffer): A rows wr*64 + ii*32 + l31 of half-tile image, B rows wc*32 + l31
    int rdA[4], rdB[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        rdA[ks] = swz_byte(wr * 64 + l31, ks * 2 + hh);
        rdB[ks] = swz_byte(wc * 32 + l31, ks * 2 + hh);
    }
    // Products are v_mfma_f32_16x16x32 (FLAGS 65536 = the r02a..r02e 32x32x16 form, kept for A/B): the same LDS images, read volume
    // and MFMA cycles, but a 16-row fragment is lane & 15 = row, lane >> 4 = which 8 of the 32 k values.  On random operands the
    // 16x16x32 form costs 11 % less energy per flop (tools/mfma_ceiling.bin: 2032 vs 1814 TFLOP/s at the 1400 W cap) and the forward
    // is energy-bound: sq8k 1276 -> 1485 TFLOP/s, ViT-B bs256 forward 10.24 -> 9.98 ms, ViT-L/384 51.4 -> 49.5 ms (profiles/r02c).
    constexpr bool M16 = (FLAGS & 65536) == 0;
    const int l15 = lane & 15, g4 = lane >> 4;
    int rdA16[2][2], rdB16[2][2];                    // [16-row tile parity][32-deep k-step]: tiles 2 apart are 32 rows = 4096 bytes apart
#pragma unroll
    for (int pz = 0; pz < 2; ++pz)
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            rdA16[pz][k2] = swz_byte(wr * 64 + pz * 16 + l15, k2 * 4 + g4);
            rdB16[pz][k2] = swz_byte(wc * 32 + pz * 16 + l15, k2 * 4 + g4);
        }
    v8 fa[2][4], fb[2][4];
    f32x16 acc[4][2];
    f32x4 acc16[8][4];
    auto read_a = [&](int buf, int h) {
        if constexpr (M16) {      // tile t (16 rows) of the half, k-step k2 -> fa[t >> 1][(t & 1) * 2 + k2]
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) fa[t >> 1][(t & 1) * 2 + k2] = *(const v8 *)(smem + off_a(buf, h) + (t >> 1) * 4096 + rdA16[t & 1][k2]);
        } else {
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) fa[ii][ks] = *(const v8 *)(smem + off_a(buf, h) + ii * 4096 + rdA[ks]);
        }
    };
    auto read_b = [&](int buf, int h) {
        if constexpr (M16) {      // tile u (16 columns) of the half, k-step k2 -> fb[h][u * 2 + k2]
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) fb[h][u * 2 + k2] = *(const v8 *)(smem + off_b(buf, h) + rdB16[u][k2]);
        } else {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) fb[h][ks] = *(const v8 *)(smem + off_b(buf, h) + rdB[ks]);
        }
    };
    // MFMAs [first, last) of a quadrant's 8 (k-step major, so consecutive MFMAs alternate between its two accumulators)
    auto mma = [&](int ha, int hb, int first, int last) {
        if constexpr (M16) {      // 16 MFMAs of 16 cycles = the same 256 cycles; k-step major, 8 independent accumulators in between
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        acc16[ha * 4 + t][hb * 2 + u] = Elem<T>::mfma16(fb[hb][u * 2 + k2], fa[t >> 1][(t & 1) * 2 + k2], acc16[ha * 4 + t][hb * 2 + u]);
        } else {
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                const int ks = n >> 1, ii = n & 1;
                if (n >= first && n < last) acc[2 * ha + ii][hb] = Elem<T>::mfma(fb[hb][ks], fa[ii][ks], acc[2 * ha + ii][hb]);
            }
        }
    };

    unsigned stamps = 0; int n_stamp = -1;          // timeline experiment: lane i of `stamps` = i-th stamp
    auto stamp = [&]() {
        if constexpr ((FLAGS & 32) != 0) {
            if (n_stamp >= 0 && n_stamp < 64) {
                const unsigned t = (unsigned)__builtin_readcyclecounter();
                stamps = lane == n_stamp ? t : stamps;
                ++n_stamp;
            }
        }
    };
    unsigned long long t_arr = 0;                   // FLAGS 1024: arrival times at the barriers (sampled before, consumed after: ~no perturbation)
    auto arrive = [&]() { if constexpr ((FLAGS & 1024) != 0) { if (n_stamp >= 0 && n_stamp < 64) t_arr = __builtin_readcyclecounter(); } };
    auto arrived = [&]() {
        if constexpr ((FLAGS & 1024) != 0) { if (n_stamp >= 0 && n_stamp < 64) { stamps = lane == n_stamp ? (unsigned)t_arr : stamps; ++n_stamp; } }
    };
    auto fine_stamp = [&]() {                       // FLAGS 64: five stamps per phase (after B2, after the stage issue, after the vmcnt wait, after B1 + reads landed, after the MFMAs)
        if constexpr ((FLAGS & 64) != 0) {
            if (n_stamp >= 0 && n_stamp < 64) {
                const unsigned t = (unsigned)__builtin_readcyclecounter();
                stamps = lane == n_stamp ? t : stamps;
                ++n_stamp;
            }
        }
    };
    // one phase = [reads, stage] | counted wait | barrier | 8 MFMAs | barrier
#define PP_PHASE(READS, STAGE, VMCNT, HA, HB, FIRST, FIRST_STMT)                                              \
    {                                                                                      \
        if constexpr ((FLAGS & (64 | 128)) != 0) {                                         \
            if (FIRST) { FIRST_STMT; }                                                     \
            if (!(FLAGS & 4)) { STAGE; }                                                   \
            fine_stamp();                                                                  \
            if (FIRST) { if (relaxed) pp_wait_vmcnt<VMCNT + 1 + pp_epi_stores<EPI>()>(); else pp_wait_vmcnt<VMCNT + 1>(); }   \
            else pp_wait_vmcnt<VMCNT>();                                                   \
            fine_stamp();                                                                  \
            if (!(FLAGS & 8)) { READS; }                                                   \
        } else {                                                                           \
            if (!(FLAGS & 8)) { READS; }                                                   \
            if (FIRST) { FIRST_STMT; }                                                     \
            if (!(FLAGS & 4)) { STAGE; }                                                   \
            if (FIRST) { if (relaxed) pp_wait_vmcnt<VMCNT + 1 + pp_epi_stores<EPI>()>(); else pp_wait_vmcnt<VMCNT + 1>(); }   \
            else pp_wait_vmcnt<VMCNT>();                                                   \
        }                                                                                  \
        arrive();                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        pp_barrier();                                                                      \
        stamp(); fine_stamp(); arrived();                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        if (!(FLAGS & 1)) __builtin_amdgcn_s_setprio(1);                                   \
        if (!(FLAGS & 16)) mma(HA, HB, 0, 8);                                              \
        if (!(FLAGS & 1)) __builtin_amdgcn_s_setprio(0);                                   \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        fine_stamp(); arrive();                                                            \
        pp_barrier();                                                                      \
        stamp(); fine_stamp(); arrived();                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                 \
    }
    // The first K-tile of a tile (`first`) also (a) zeroes each accumulator quadrant in the load part of the phase that first uses
    // it (the wave is waiting for its partner there anyway), (b) DMA-loads the bias of the wave's 64 columns into its epilogue
    // patch (one more vector-memory op, issued BEFORE the phase's stage so only this K-tile's four waits count it), and
    // (c) if a full-tile epilogue ran just before (`relaxed`), skips over exactly pp_epi_stores() stores in those waits: they are
    // younger than the stages being retired, and draining them costs microseconds when every CU stores at once.
    bool relaxed = false;
    int bias_so = 0;                                 // byte offset of the consumer tile's bias columns
    __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc((void *)g.bias, 0, (int)0xffffffffu, 0x00020000);
    auto zero_quadrant = [&](int ha, int hb) {
        if constexpr (M16) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u) acc16[ha * 4 + t][hb * 2 + u] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        } else {
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[2 * ha + ii][hb][r] = 0.0f;
        }
    };
    auto stage_bias = [&]() { __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, LPTR(smem + LDS + wave * 4096), 4, lane * 4, bias_so, 0, 0); };
    auto ktile = [&](auto bc, bool first) {
        constexpr int B = decltype(bc)::value;       // buffer of the K-tile being consumed
        constexpr int W8 = LEAD * STAGE_OPS;
        const bool f = B == 0 && first;
        PP_PHASE((read_a(B, 0), read_b(B, 0)), stage_w(1, off_b(B ^ 1, 1)), W8, 0, 0, f, (zero_quadrant(0, 0), stage_bias()))   // C00 ; B1 of the next K-tile
        PP_PHASE(read_b(B, 1), (stage_a(1, off_a(B ^ 1, 1)), advance()), W8, 0, 1, f, zero_quadrant(0, 1))                        // C01 ; A1 of the next K-tile
        PP_PHASE(read_a(B, 1), stage_a(0, off_a(B, 0)), W8, 1, 1, f, zero_quadrant(1, 1))                                     // C11 ; A0 two K-tiles ahead
        PP_PHASE((void)0, stage_w(0, off_b(B, 0)), W8, 1, 0, f, zero_quadrant(1, 0))                                          // C10 ; B0 two K-tiles ahead
        if (f) relaxed = false;
    };
    // ---- two-burst schedule (FLAGS & 4096): the same operand stream, fragments and MFMA order with HALF the barriers.
    // A K-tile is two bursts of 16 MFMAs: Q0 = C00 + C01 (reads A0, B0, B1), Q1 = C11 + C10 (reads A1; B0/B1 stay in registers).
    // Stream order is unchanged (A0 B0 B1 A1 per K-tile): Q0 stages A1 of the next K-tile, Q1 stages A0 B0 B1 two K-tiles ahead
    // (into the buffer whose A0/B0/B1 slots this K-tile's Q0 just read), so every stage has two bursts to land and
    // s_waitcnt vmcnt(8) is uniform; the wait also drains this wave's LDS reads (see pp_wait_vm_lgkm).
#define PP_BURST(READS, STAGE, FIRST, FIRST_STMT, MMA)                                                        \
    {                                                                                      \
        if (!(FLAGS & 8)) { READS; }                                                       \
        if (FIRST) { FIRST_STMT; }                                                         \
        if (!(FLAGS & 4)) { STAGE; }                                                       \
        if (FIRST) { if (relaxed) pp_wait_vm_lgkm<W8B + 1 + pp_epi_stores<EPI>()>(); else pp_wait_vm_lgkm<W8B + 1>(); }   \
        else pp_wait_vm_lgkm<W8B>();                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        pp_barrier();                                                                      \
        stamp();                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        if (!(FLAGS & 1)) __builtin_amdgcn_s_setprio(1);                                   \
        if (!(FLAGS & 16)) { MMA; }                                                        \
        if (!(FLAGS & 1)) __builtin_amdgcn_s_setprio(0);                                   \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        pp_barrier();                                                                      \
        stamp();                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                 \
    }
    auto ktile2 = [&](auto bc, bool first) {
        constexpr int B = decltype(bc)::value;
        constexpr int W8B = 4 * STAGE_OPS;           // A1 of the next K-tile + A0 B0 B1 two K-tiles ahead may stay in flight
        const bool f = B == 0 && first;
        PP_BURST((read_a(B, 0), read_b(B, 0), read_b(B, 1)), (stage_a(1, off_a(B ^ 1, 1)), advance()), f,
                 (zero_quadrant(0, 0), zero_quadrant(0, 1), stage_bias()), (mma(0, 0, 0, 8), mma(0, 1, 0, 8)))
        PP_BURST(read_a(B, 1), (stage_a(0, off_a(B, 0)), stage_w(0, off_b(B, 0)), stage_w(1, off_b(B, 1))), f,
                 (zero_quadrant(1, 1), zero_quadrant(1, 0)), (mma(1, 1, 0, 8), mma(1, 0, 0, 8)))
        if (f) relaxed = false;
    };
    typedef std::integral_constant<int, 0> I0; typedef std::integral_constant<int, 1> I1;

    // ---- prologue: A0 B0 B1 A1 of K-tile 0 and A0 B0 of K-tile 1 in flight, the first two landed
    stage_a(0, off_a(0, 0)); stage_w(0, off_b(0, 0)); stage_w(1, off_b(0, 1)); stage_a(1, off_a(0, 1)); advance();     // nkt >= 2: K-tile 1 exists
    stage_a(0, off_a(1, 0)); stage_w(0, off_b(1, 0));
    if constexpr ((FLAGS & 4096) != 0) stage_w(1, off_b(1, 1));     // two-burst schedule: A0 B0 B1 of K-tile 1 precede its Q0; K-tile 0 complete but A1
    pp_wait_vmcnt<4 * STAGE_OPS>();
    pp_barrier();
    if (!(FLAGS & 2) && wr == 1) pp_barrier();      // the second wave row runs one barrier behind the first

    for (int round = 0; round < my_tiles; ++round) {
        int m0, n0; tile_origin(round, m0, n0);
        bias_so = __builtin_amdgcn_readfirstlane((n0 + wc * 64) * 4);
        for (int kt = 0; kt < nkt; kt += 2) {
            if constexpr ((FLAGS & (32 | 64 | 1024)) != 0) { if (round == 0 && kt == 4) n_stamp = 0; }
            if constexpr ((FLAGS & 4096) != 0) { ktile2(I0{}, kt == 0); ktile2(I1{}, false); }
            else { ktile(I0{}, kt == 0); ktile(I1{}, false); }
        }
        if constexpr ((FLAGS & 8) != 0) {           // fragments never read: keep the MFMA operands "defined" for the compiler
            if (round == 0) { for (int ii = 0; ii < 2; ++ii) for (int ks = 0; ks < 4; ++ks) { asm volatile("" : "+v"(fa[ii][ks])); asm volatile("" : "+v"(fb[ii][ks])); } }
        }
        if constexpr ((FLAGS & 16) != 0) { for (int ii = 0; ii < 2; ++ii) for (int ks = 0; ks < 4; ++ks) { asm volatile("" :: "v"(fa[ii][ks]), "v"(fb[ii][ks])); } }
        const bool full = (m0 + BM <= g.M_real) && (n0 + BN <= g.N);
        // The second wave row runs one barrier behind, so its last K-loop barrier would only be released by the first row's first
        // barrier of the NEXT tile -- i.e. after the first row's epilogue, and the two rows' epilogues would run one after the other.
        // Aligning the rows here (and restoring the offset after the epilogue) lets both epilogues run at the same time: forward
        // 10.82 -> 10.75 ms, fc1 +4 % (r02c; FLAGS 8192 = the unaligned r02a behaviour, kept for A/B).
        if constexpr ((FLAGS & 8192) == 0) { if (!(FLAGS & 2) && wr == 0) pp_barrier(); }
        if constexpr (M16) {
            if constexpr ((FLAGS & 2048) != 0) {
#pragma unroll
                for (int t = 0; t < 8; ++t) asm volatile("" :: "v"(acc16[t][0]), "v"(acc16[t][1]), "v"(acc16[t][2]), "v"(acc16[t][3]));
            } else if (full && EPI != EPI_PATCH && !(FLAGS & 512)) {
                constexpr int esz = (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) ? 2 : 4;
                const int voff = ((wr * 128 + (lane >> 3)) * g.ldo + wc * 64) * esz + (lane & 7) * 16;
                f32x4 bq[4];                       // bias of columns u * 16 + 4 g4 .. + 3, staged into the wave's patch by LDS-DMA during the K loop
#pragma unroll
                for (int u = 0; u < 4; ++u) bq[u] = *(const f32x4 *)(smem + LDS + wave * 4096 + u * 64 + g4 * 16);
                epilogue16_staged<T, EPI, 4, (FLAGS & 32768) ? 16 : 0>(acc16, bq, rsrcO, smem + LDS + wave * 4096, voff, __builtin_amdgcn_readfirstlane((m0 * g.ldo + n0) * esz), 8 * g.ldo * esz, lane);
                relaxed = true;
            } else {
                const int row0 = m0 + wr * 128 + l15, ncol = n0 + wc * 64;
                if (full) epilogue16<T, EPI, 8, 4, true>(g, acc16, row0, ncol + 4 * g4);
                else epilogue16<T, EPI, 8, 4, false>(g, acc16, row0, ncol + 4 * g4);
            }
        } else if constexpr ((FLAGS & 2048) != 0) {
            asm volatile("" :: "v"(acc[0][0]), "v"(acc[1][0]), "v"(acc[2][0]), "v"(acc[3][0]), "v"(acc[0][1]), "v"(acc[1][1]), "v"(acc[2][1]), "v"(acc[3][1]));
        } else if (full && EPI != EPI_PATCH && !(FLAGS & 512)) {
            constexpr int esz = (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) ? 2 : 4;
            // row layout of the stores: lane -> row (lane>>3) + 8t of a 32-row block, 16-byte piece lane&7 of the wave's 128-byte row segment
            const int voff = ((wr * 128 + (lane >> 3)) * g.ldo + wc * 64) * esz + (lane & 7) * 16;
            // readfirstlane: the tile origin comes out of an integer division done on the VALU; without it hipcc wraps every
            // buffer op in a waterfall loop over the (uniform) SGPR offset
            pp_epilogue_full<T, EPI, (FLAGS & 16384) != 0, (FLAGS & 32768) ? 16 : 0>(acc, rsrcO, smem + LDS + wave * 4096, voff, __builtin_amdgcn_readfirstlane((m0 * g.ldo + n0) * esz), 8 * g.ldo * esz, lane);
            relaxed = true;
        } else {
            const int row0 = m0 + wr * 128 + l31, ncol = n0 + wc * 64;
            if (full) pp_epilogue<T, EPI, true>(g, acc, row0, ncol, hh);
            else pp_epilogue<T, EPI, false>(g, acc, row0, ncol, hh);
        }
        if constexpr ((FLAGS & 32768) != 0) {
            // ---- LayerNorm of row blocks whose last column tile just finished (EPI_BIAS_RESID, N == ldo == hidden size).
            // The wave rows are aligned here.  Hand-off between workgroups (cdna_hip_programming.md Guideline 16, fan-in form): the tile's
            // X stores were write-through (sc1) -> every wave drains them -> barrier -> ONE lane takes a ticket (relaxed, agent scope);
            // the workgroup that draws the last ticket of the row block acquires (drops its stale L1/L2 lines) and normalises the
            // block's rows exactly as layernorm_kernel does (same sums in the same order): bit-identical U.
            if (!(g.dbg & 128)) {
            pp_wait_vmcnt<0>();
            pp_barrier();
            volatile int *flag = (volatile int *)(smem + LDS + 2048);       // wave 0's patch, beyond the bias bytes
            if (tid == 0) {
                if (!full) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }   // ragged tile: plain stores
                int *cnt = g.ln_cnt + m0 / BM;
                const int old = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int last = old == ntn - 1;
                if (last) { __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
                *flag = last;
            }
            pp_barrier();
            if (*flag && !(g.dbg & 64)) pp_layernorm_rows<T>(g, m0, min(BM, g.M_real - m0), wave, lane);
            relaxed = false;                       // everything was drained above
            }
        }
        if constexpr ((FLAGS & 8192) == 0) { if (!(FLAGS & 2) && wr == 1) pp_barrier(); }
    }
    if (!(FLAGS & 2) && wr == 0) pp_barrier();
    pp_wait_vmcnt<0>();                             // the trailing (unused) stages must land before the LDS allocation is released
    if constexpr ((FLAGS & (32 | 64 | 1024)) != 0) ((unsigned *)g.pos)[((size_t)bid * 8 + wave) * 64 + lane] = stamps;
#undef PP_PHASE
#undef PP_BURST
}

bool gemm_pp_supports(const GemmArgs &a) {
    // byte offsets into A, W and out are 32-bit (buffer addressing)
    const size_t lim = 0xf0000000u;
    if ((size_t)a.M * a.lda * 2 > lim || (size_t)a.N_pad * a.ldw * 2 > lim || (size_t)(a.M + a.M / 64 + 2) * a.ldo * 4 > lim) return false;
    return a.M % pp::BM == 0 && a.N_pad % pp::BN == 0 && a.K % (2 * pp::BK) == 0 && a.K >= 2 * pp::BK && a.N % 4 == 0 && a.ldo % 4 == 0;
}

template <typename T, int EPI, int FLAGS>
static hipError_t launch_pp_inst(const GemmArgs &a, int n_cu, hipStream_t stream, bool prepare) {
    if (prepare || FLAGS) {     // once per device (tuning_for_device); the experiment builds set it on every launch
        hipError_t e = hipFuncSetAttribute((const void *)gemm_pp_kernel<T, EPI, FLAGS>, hipFuncAttributeMaxDynamicSharedMemorySize, pp::LDS_ALL);
        if (prepare) return e;
    }
    const int ntiles = (a.M / pp::BM) * (a.N_pad / pp::BN);
    int cap = n_cu & ~7;                             // the tile walk keeps a workgroup on one XCD: grid is a multiple of 8
    if (cap <= 0) cap = 256;
    const int grid = ntiles < cap ? ntiles : cap;
    hipLaunchKernelGGL((gemm_pp_kernel<T, EPI, FLAGS>), dim3(grid), dim3(512), pp::LDS_ALL, stream, a);
    return hipGetLastError();
}
template <typename T>
static hipError_t launch_pp_t(int epi, const GemmArgs &a, int n_cu, hipStream_t stream, int flags, bool prepare) {
    if (flags == 32768) {      // LayerNorm of finished row blocks fused into the residual GEMM
        if (epi != EPI_BIAS_RESID || !a.ln_out || !a.ln_cnt || !a.ln_w || !a.ln_b || a.N != a.ldo || a.N % 256 || a.N > 1536) return hipErrorInvalidValue;
        return launch_pp_inst<T, EPI_BIAS_RESID, 32768>(a, n_cu, stream, prepare);
    }
    if (flags == 8192) {       // wave rows NOT aligned at the epilogue (the r02a behaviour, for A/B): every epilogue
        switch (epi) {
        case EPI_BIAS: return launch_pp_inst<T, EPI_BIAS, 8192>(a, n_cu, stream, prepare);
        case EPI_BIAS_GELU: return launch_pp_inst<T, EPI_BIAS_GELU, 8192>(a, n_cu, stream, prepare);
        case EPI_BIAS_RESID: return launch_pp_inst<T, EPI_BIAS_RESID, 8192>(a, n_cu, stream, prepare);
        case EPI_BIAS_F32: return launch_pp_inst<T, EPI_BIAS_F32, 8192>(a, n_cu, stream, prepare);
        case EPI_PATCH: return launch_pp_inst<T, EPI_PATCH, 8192>(a, n_cu, stream, prepare);
        default: return hipErrorInvalidValue;
        }
    }
    if (flags == 65536) {      // v_mfma_32x32x16 products (the r02a..r02e kernel): every epilogue
        switch (epi) {
        case EPI_BIAS: return launch_pp_inst<T, EPI_BIAS, 65536>(a, n_cu, stream, prepare);
        case EPI_BIAS_GELU: return launch_pp_inst<T, EPI_BIAS_GELU, 65536>(a, n_cu, stream, prepare);
        case EPI_BIAS_RESID: return launch_pp_inst<T, EPI_BIAS_RESID, 65536>(a, n_cu, stream, prepare);
        case EPI_BIAS_F32: return launch_pp_inst<T, EPI_BIAS_F32, 65536>(a, n_cu, stream, prepare);
        case EPI_PATCH: return launch_pp_inst<T, EPI_PATCH, 65536>(a, n_cu, stream, prepare);
        default: return hipErrorInvalidValue;
        }
    }
    if (flags == 4096) {       // two-burst schedule: every epilogue
        switch (epi) {
        case EPI_BIAS: return launch_pp_inst<T, EPI_BIAS, 4096>(a, n_cu, stream, prepare);
        case EPI_BIAS_GELU: return launch_pp_inst<T, EPI_BIAS_GELU, 4096>(a, n_cu, stream, prepare);
        case EPI_BIAS_RESID: return launch_pp_inst<T, EPI_BIAS_RESID, 4096>(a, n_cu, stream, prepare);
        case EPI_BIAS_F32: return launch_pp_inst<T, EPI_BIAS_F32, 4096>(a, n_cu, stream, prepare);
        case EPI_PATCH: return launch_pp_inst<T, EPI_PATCH, 4096>(a, n_cu, stream, prepare);
        default: return hipErrorInvalidValue;
        }
    }
    if (flags) {       // experiment builds exist for the plain bias epilogue only
        if (epi != EPI_BIAS) return hipErrorInvalidValue;
        switch (flags) {
        case 1: return launch_pp_inst<T, EPI_BIAS, 1>(a, n_cu, stream, prepare);
        case 2: return launch_pp_inst<T, EPI_BIAS, 2>(a, n_cu, stream, prepare);
        case 4: return launch_pp_inst<T, EPI_BIAS, 4>(a, n_cu, stream, prepare);
        case 8: return launch_pp_inst<T, EPI_BIAS, 8>(a, n_cu, stream, prepare);
        case 12: return launch_pp_inst<T, EPI_BIAS, 12>(a, n_cu, stream, prepare);
        case 16: return launch_pp_inst<T, EPI_BIAS, 16>(a, n_cu, stream, prepare);
        case 20: return launch_pp_inst<T, EPI_BIAS, 20>(a, n_cu, stream, prepare);
        case 24: return launch_pp_inst<T, EPI_BIAS, 24>(a, n_cu, stream, prepare);
        case 32: return launch_pp_inst<T, EPI_BIAS, 32>(a, n_cu, stream, prepare);
        case 36: return launch_pp_inst<T, EPI_BIAS, 36>(a, n_cu, stream, prepare);
        case 40: return launch_pp_inst<T, EPI_BIAS, 40>(a, n_cu, stream, prepare);
        case 44: return launch_pp_inst<T, EPI_BIAS, 44>(a, n_cu, stream, prepare);
        case 56: return launch_pp_inst<T, EPI_BIAS, 56>(a, n_cu, stream, prepare);
        case 64: return launch_pp_inst<T, EPI_BIAS, 64>(a, n_cu, stream, prepare);
        case 128: return launch_pp_inst<T, EPI_BIAS, 128>(a, n_cu, stream, prepare);
        case 256: return launch_pp_inst<T, EPI_BIAS, 256>(a, n_cu, stream, prepare);
        case 512: return launch_pp_inst<T, EPI_BIAS, 512>(a, n_cu, stream, prepare);
        case 1024: return launch_pp_inst<T, EPI_BIAS, 1024>(a, n_cu, stream, prepare);
        case 2048: return launch_pp_inst<T, EPI_BIAS, 2048>(a, n_cu, stream, prepare);
        case 16384: return launch_pp_inst<T, EPI_BIAS, 16384>(a, n_cu, stream, prepare);
        case 1028: return launch_pp_inst<T, EPI_BIAS, 1028>(a, n_cu, stream, prepare);
        case 1032: return launch_pp_inst<T, EPI_BIAS, 1032>(a, n_cu, stream, prepare);
        case 1036: return launch_pp_inst<T, EPI_BIAS, 1036>(a, n_cu, stream, prepare);
        case 4097: return launch_pp_inst<T, EPI_BIAS, 4097>(a, n_cu, stream, prepare);
        case 4100: return launch_pp_inst<T, EPI_BIAS, 4100>(a, n_cu, stream, prepare);
        case 4104: return launch_pp_inst<T, EPI_BIAS, 4104>(a, n_cu, stream, prepare);
        case 4108: return launch_pp_inst<T, EPI_BIAS, 4108>(a, n_cu, stream, prepare);
        case 4128: return launch_pp_inst<T, EPI_BIAS, 4128>(a, n_cu, stream, prepare);
        case 4140: return launch_pp_inst<T, EPI_BIAS, 4140>(a, n_cu, stream, prepare);
        case 6144: return launch_pp_inst<T, EPI_BIAS, 6144>(a, n_cu, stream, prepare);
        default: return hipErrorInvalidValue;
        }
    }
    switch (epi) {
    case EPI_BIAS: return launch_pp_inst<T, EPI_BIAS, 0>(a, n_cu, stream, prepare);
    case EPI_BIAS_GELU: return launch_pp_inst<T, EPI_BIAS_GELU, 0>(a, n_cu, stream, prepare);
    case EPI_BIAS_RESID: return launch_pp_inst<T, EPI_BIAS_RESID, 0>(a, n_cu, stream, prepare);
    case EPI_BIAS_F32: return launch_pp_inst<T, EPI_BIAS_F32, 0>(a, n_cu, stream, prepare);
    case EPI_PATCH: return launch_pp_inst<T, EPI_PATCH, 0>(a, n_cu, stream, prepare);
    default: return hipErrorInvalidValue;
    }
}
hipError_t launch_gemm_pp(int dtype, int epi, const GemmArgs &a, int n_cu, hipStream_t stream, int flags, bool prepare) {
    if (!prepare && !gemm_pp_supports(a)) return hipErrorInvalidValue;
    return dtype == DT_F16 ? launch_pp_t<_Float16>(epi, a, n_cu, stream, flags, prepare) : launch_pp_t<__bf16>(epi, a, n_cu, stream, flags, prepare);
}

}  // namespace vitx

// gemm_pp.hip -- "ping-pong" persistent MFMA GEMM for gfx950: the wide-tile kernel of the ViT forward path.
//
//   C[M][N] = A[M][K] . W[N][K]^T  (+ fused epilogue), A/W fp16 or bf16, f32 accumulate
//   (ggml_mul_mat at /root/reference/vit.cpp:772,820,868,889,896 with the bias / GELU / residual / pos-embed ops fused).
//
// Structure (DESIGN.md "GEMM"):
//   * 512 threads = 8 waves as 2(M) x 4(N); tile 256x256, BK = 64; each wave owns 128x64 of C as 8x4 accumulators of
//     v_mfma_f32_16x16x32 (the 4x2 v_mfma_f32_32x32x16 form of r02a-e costs the same cycles and 11 % more energy per flop:
//     profiles/r02f).  One persistent workgroup per CU walks its tiles and keeps ONE operand stream running across tile boundaries.
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
// Cache-policy bits (buffer aux: 2 = nt) of the tensors one launch writes and ONE later launch reads once: QKV (qkv -> attention) and the MLP
// hidden tensor H (fc1 -> fc2) are stored and fetched non-temporally, so that they stream through the L2s / the Infinity Cache instead of
// evicting what is re-read: the residual stream X, the normalised rows and the weights -- also those of the OTHER sub-batch stream's kernel.
// r05 A/B, six interleaved rounds on one box (profiles/r05/ab_cache_policy.txt): 9.705 ms default policy everywhere, H alone 9.63, QKV alone
// 9.69, both 9.55 / 9.48 (+1.7 % images/s; every kernel is a little SLOWER alone -- 11.0 -> 11.7 ms of exclusive time -- the gain is what the
// two streams stop taking from each other).  nt on the A streams of qkv / fc1 (the normalised rows, which nine / twelve column tiles
// re-read from L2) costs 6 %: PP_AUX_A stays 0.
#ifndef PP_AUX_QKV
#define PP_AUX_QKV 2            // stores of the EPI_BIAS / EPI_BIAS_HILO epilogue (qkv -> attention)
#endif
#ifndef PP_AUX_H
#define PP_AUX_H 2              // stores of the EPI_BIAS_GELU epilogue (fc1 -> fc2)
#endif
#ifndef PP_AUX_A_LNF
#define PP_AUX_A_LNF 2          // A-operand LDS-DMA of the LayerNorm-fusing launches (proj: the attention output, fc2: H)
#endif
#ifndef PP_AUX_A
#define PP_AUX_A 0              // A-operand LDS-DMA of every other launch (qkv, fc1: the normalised rows)
#endif
#ifndef PP_RESID_DEPTH
#define PP_RESID_DEPTH 1        // passes the residual rows of the LayerNorm-fusing epilogue are requested ahead (r04 A/B: 1 = 3 = 5 = 7, profiles/r04/ab_resid_depth.txt)
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
constexpr int GROUP_M = 8;              // raster group height (row blocks) when the launcher does not choose one (GemmArgs::group_m)
constexpr int STAGE_OPS = 2;            // LDS-DMA instructions per thread per half-tile
constexpr int LEAD = 4;                 // stages allowed in flight past a phase's wait
}  // namespace pp

template <int N> __device__ __forceinline__ void pp_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// counted vector-memory wait + every LDS read of this wave landed (the two-burst schedule re-stages a half-tile one burst after
// its last read: the reads must be complete BEFORE the barrier that releases the other wave row's stage issue)
template <int N> __device__ __forceinline__ void pp_wait_vm_lgkm() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void pp_barrier() { asm volatile("s_barrier" ::: "memory"); }

// whole-row stores of the staged epilogue per wave and tile (epilogue16_staged): the next tile's K loop skips over exactly this many
template <int EPI> __host__ __device__ constexpr int pp_epi_stores() { return (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) ? 16 : 32; }      // (EPI_BIAS_HILO: two 16-bit planes = 32)

// ---- EPI_BIAS_RESID + LayerNorm of the finished rows (GemmLn, kernels.h), for one full 256 x 256 tile of the persistent kernel.
// Statistics follow device_common.h "LayerNorm statistics by 256-column tiles": this workgroup's tile is tile c = n0 / 256 of its rows;
// wave column w = the 64-column chunk, accumulator half j and 16-byte piece k = lane & 7 of the staged row layout name the pieces.
//   passes   the staged f32 epilogue (bias, + residual; 16 passes of 16 rows x 32 columns) into row layout; the final values stay in the
//            accumulator registers and are stored as whole lines at the very end
//   (1)      tile sums: in-lane over j, ln_sum8 over k, the four wave columns through LDS            -> mean_c   [2 barriers]
//   (2)      centred sums of squares the same way                                                    -> M2_c     [1 barrier]
//   (3)      one thread per row publishes {mean_c, M2_c} as two 8-byte granules {value, tag = epoch} (ONE sc1 store each: the data
//            is the flag, cdna_hip_programming.md Guideline 16 R2) and polls the other column tiles' granules of the same rows
//            (relaxed agent-scope loads, s_sleep, bounded by the wall clock)                                     [1 barrier]
//   (4)      ln_combine in tile order; every lane normalises the 16 rows x 8 columns it holds and stores them to ln.out  [1 barrier]
// LDS: the statistics live in the operand ring's half-tile slot A1 of buffer 1 -- last read in the tile's final K-tile, re-staged only
// in phase 2 of the next tile's first K-tile, and every wave is past the K loop here (the wave rows are aligned around the epilogue).
// Returns true when exactly pp_epi_stores<EPI_BIAS_RESID>() = 32 stores per lane are still in flight (the caller's counted vmcnt
// skips them), false when the tile was left to the fix-up (everything drained).
// In-place safety (proj: A == an earlier LayerNorm's output buffer is NOT this one: the engine alternates two U buffers).
template <typename T>
__device__ __forceinline__ bool pp_epilogue_ln(const GemmArgs &g, const GemmLn &ln, f32x4 (&acc)[8][4], __amdgpu_buffer_rsrc_t ro, char *smem,
                                               int voff_in, int soff, int soff8, int tid_in, int m0, int n0, int ntn) {
    using namespace pp;
    typedef unsigned long long u64;
    // Everything per-lane below is derived from OPAQUE copies made here, once per tile: the addresses are invariant across the persistent
    // tile loop, and hipcc otherwise hoists some 80 of them in front of the K loops -- where the kernel has no register to spare -- and spills them
    // (a spill is a vector-memory operation: it would also break the counted vmcnt of the K loop).
    int tid = tid_in, voff = voff_in;
    asm volatile("" : "+v"(tid), "+v"(voff));
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wr = wave >> 2, wc = wave & 3;
    const int l15 = lane & 15, g4 = lane >> 4, lr = lane >> 3, lk = lane & 7;
    char *patch = smem + LDS + wave * 4096;
    char *sb = smem + off_a(1, 1);                            // 16 KiB free during the epilogue (see above)
    float *part = (float *)sb;                                // [4 wave columns][256 rows]
    float *cmean = (float *)(sb + 4096);                      // [256] mean of this tile's 256 columns
    float *stat = (float *)(sb + 5120);                       // [LN_MAX_TILES][256][2] {mean_c, M2_c} of every column tile of these rows
    float *fin = (float *)(sb + 13312);                       // [256][2] {mean, rstd}
    int *fail = (int *)(sb + 15360);
    const int rd_off = lr * 128 + ((lk ^ (lr & 7)) * 16);     // row layout of the patch: row lr + 8 t, 16-byte piece lk
    const int mb = m0 / BM, ct = n0 / BN;
    // ---- passes (epilogue16_staged, f32 + residual), keeping the stored values: xv(c, t) = row (c >> 1) * 32 + lr + 8 t, columns (c & 1) * 32 + 4 lk ..
    // (register budget: the kernel sits at 256.  The bias is therefore added in ROW layout -- 2 x 4 values per lane instead of the 4 x 4 of
    // the accumulator layout, same (acc + bias) + x order -- and nothing but xv accumulates across the passes.)
    // xv(b, j, t) ALIASES the accumulator registers pass (b, j) has just consumed: for hipcc the accumulators stay live into the next tile's
    // K loop (they are re-zeroed under a run-time condition), so a second 128-register array could only be spilled.
    // xv(b, j, t) = row 16 b + lr + 8 t of the wave's 128, columns 32 j + 4 lk .. + 3 of its 64.
#define xv(b, j, t) acc[b][2 * (j) + (t)]
    {
        f32x4 br[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) br[j] = *(const f32x4 *)(patch + j * 128 + lk * 16);       // bias of columns j * 32 + 4 lk .. of the wave's 64 (LDS-DMA'd during the K loop)
        pp_lds_fence();
        const int wr_row = l15 * 128, x16 = (l15 & 7) * 16;
        // residual rows: loaded RD passes ahead of their use.  r04 tested the hypothesis that one pass ahead (16 KiB in flight per CU) leaves
        // the epilogue latency-bound: depths 1 / 3 / 5 / 7 run the forward in 9.55 / 9.67 / 9.64 / 9.56 ms (interleaved, one box) -- it is not;
        // all 240 workgroups of a round are in their epilogue at the same time and together they move 640 KB per tile at ~4.5 TB/s
        constexpr int RD = PP_RESID_DEPTH;
        u32x4 res[RD + 1][2];
        auto load_res = [&](int c, u32x4 (&dst)[2]) {
            const int b = c >> 1, j = c & 1;
#pragma unroll
            for (int t = 0; t < 2; ++t) dst[t] = __builtin_amdgcn_raw_buffer_load_b128(ro, voff + j * 128, soff + (b * 2 + t) * soff8, 0);
        };
        auto write_pass = [&](int c) {          // pass c = (16-row block c >> 1, 32-column half c & 1) into half c & 1 of the patch (epilogue16_staged)
            const int b = c >> 1, j = c & 1;
            char *pb = patch + (c & 1) * 2048 + wr_row;
#pragma unroll
            for (int uu = 0; uu < 2; ++uu) *(f32x4 *)(pb + (((4 * uu + g4) * 16) ^ x16)) = acc[b][2 * j + uu];
            pp_lds_fence();
        };
#pragma unroll
        for (int c = 0; c < RD; ++c) load_res(c, res[c % (RD + 1)]);
        write_pass(0);
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int b = c >> 1, j = c & 1;
            if (c + RD < 16) load_res(c + RD, res[(c + RD) % (RD + 1)]);
            f32x4 d[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) d[t] = *(const f32x4 *)(patch + (c & 1) * 2048 + t * 1024 + rd_off);
            pp_lds_fence();
            if (c + 1 < 16) write_pass(c + 1);          // reads the accumulators of pass c + 1: other registers than xv(b, j, .) below
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                xv(b, j, t) = (d[t] + br[j]) + __builtin_bit_cast(f32x4, res[c % (RD + 1)][t]);     // (acc + bias) + x, the reference's order (vit.cpp:868-873)
                asm volatile("" : "+v"(xv(b, j, t)));      // materialise NOW: with its first use far below, LLVM sinks the add and keeps (spills) all 32 residual loads
            }
        }
    }
    // LayerNorm weight / bias of this lane's 2 x 4 columns: in flight under the statistics
    f32x4 gw[2], gb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) { gw[j] = *(const f32x4 *)(ln.w + n0 + wc * 64 + j * 32 + lk * 4); gb[j] = *(const f32x4 *)(ln.b + n0 + wc * 64 + j * 32 + lk * 4); }
    // The X tile is stored AFTER the statistics exchange (r03c): stores issued before it sat in front of the granule stores and of the
    // polling loads in this CU's (in-order) vector-memory queue -- the hand-off then cost the drain of 256 KiB per workgroup on both sides.
    auto store_x = [&]() {
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int b = c >> 1, j = c & 1;
#pragma unroll
            for (int t = 0; t < 2; ++t) pp_store_b128(__builtin_bit_cast(u32x4, xv(b, j, t)), ro, voff + j * 128, soff + (b * 2 + t) * soff8);
        }
    };
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    if (tid == 0) *fail = 0;
    // ---- (1) tile sums -> mean_c
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float p = ln_sum8(ln_piece_sum(xv(b, 0, t)) + ln_piece_sum(xv(b, 1, t)));
            if (lk == 0) part[wc * 256 + wr * 128 + b * 16 + t * 8 + lr] = p;
            __builtin_amdgcn_sched_barrier(0);      // one row at a time: hipcc otherwise interleaves all 16 and spills the tile it is reducing
        }
    lds_barrier();
    if (tid < 256) cmean[tid] = (((part[tid] + part[256 + tid]) + part[512 + tid]) + part[768 + tid]) * (1.0f / 256.0f);
    lds_barrier();
    // ---- (2) centred sums of squares -> M2_c
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float mc = cmean[wr * 128 + b * 16 + t * 8 + lr];
            const float q = ln_sum8(ln_piece_sq(xv(b, 0, t), mc) + ln_piece_sq(xv(b, 1, t), mc));
            if (lk == 0) part[wc * 256 + wr * 128 + b * 16 + t * 8 + lr] = q;        // every wave read cmean, not part, since the last barrier
            __builtin_amdgcn_sched_barrier(0);
        }
    lds_barrier();
    // ---- (3) publish this tile's statistics, collect the peers'
    const unsigned epoch = ln.epoch;
    const bool fake = (ln.test & 1) && ((mb * ntn + ct) % 5 == 0);          // parity tests: this tile behaves as if a peer had timed out
    {
        const int r = tid & 255;
        if (tid < 256) {
            const float m = cmean[r], q2 = ((part[r] + part[256 + r]) + part[512 + r]) + part[768 + r];
            stat[(ct * 256 + r) * 2] = m; stat[(ct * 256 + r) * 2 + 1] = q2;
            if (!(fake && (ln.test & 2))) {
                u64 *gp = ln.sync + ((size_t)(mb * ntn + ct) * 256 + r) * 2;
                __hip_atomic_store(gp, ((u64)epoch << 32) | __builtin_bit_cast(unsigned, m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(gp + 1, ((u64)epoch << 32) | __builtin_bit_cast(unsigned, q2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        for (int pi = tid >> 8; pi < ntn - 1; pi += 2) {                   // the two halves of the workgroup poll different peers
            const int c2 = pi < ct ? pi : pi + 1;
            const u64 *gp = ln.sync + ((size_t)(mb * ntn + c2) * 256 + r) * 2;
            const long long t_start = wall_clock64();
            bool ok = false;
            for (;;) {
                const u64 a = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), b = __hip_atomic_load(gp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)(a >> 32) == epoch && (unsigned)(b >> 32) == epoch) {
                    stat[(c2 * 256 + r) * 2] = __builtin_bit_cast(float, (unsigned)a); stat[(c2 * 256 + r) * 2 + 1] = __builtin_bit_cast(float, (unsigned)b);
                    ok = true; break;
                }
                if (wall_clock64() - t_start > (long long)ln.timeout) break;
                __builtin_amdgcn_s_sleep(8);
            }
            if (!ok) *fail = 1;
        }
        if (fake && tid == 0) *fail = 1;
    }
    lds_barrier();
    if (*fail) {           // uniform: leave the row block to launch_layernorm_fixup (it recomputes every tile of the block from X: same bits)
        if (tid == 0) {
            __hip_atomic_store(ln.todo + mb, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(ln.fallbacks, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        lds_barrier();     // `fail` and the statistics are re-used by the next tile: nobody may still be reading them
        store_x();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return false;
    }
    // ---- (4) whole-row statistics, normalise this tile from registers
    if (tid < 256) {
        float m_[LN_MAX_TILES], q_[LN_MAX_TILES];
#pragma unroll
        for (int c = 0; c < LN_MAX_TILES; ++c) { m_[c] = c < ntn ? stat[(c * 256 + tid) * 2] : 0.0f; q_[c] = c < ntn ? stat[(c * 256 + tid) * 2 + 1] : 0.0f; }
        float mean, rstd;
        ln_combine(m_, q_, ntn, g.N, ln.eps, mean, rstd);
        fin[tid * 2] = mean; fin[tid * 2 + 1] = rstd;
    }
    lds_barrier();
    {
        const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(ln.out, 0, (int)0xffffffffu, 0x00020000);
        // ln.out is [M][N] of T with the GEMM's row length: the X offsets halve (f32 -> 16 bit)
        const int uoff = voff >> 1, usoff = soff >> 1, usoff8 = soff8 >> 1;
        // A lane holds columns 4 lk .. + 3 of BOTH 32-column halves; neighbouring lanes (lk ^ 1) swap one half so that the even lane owns 8
        // consecutive columns of half 0 and the odd lane 8 of half 1: ONE 16-byte store per lane and row, whole 128-byte lines per
        // instruction (8 rows), instead of two 8-byte stores to half lines.
        const bool odd = lk & 1;
        const int ucol = odd ? 64 + (lk - 1) * 8 : lk * 8;          // byte offset inside the wave's 128-byte row segment
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const f32x2 mr = *(const f32x2 *)(fin + (wr * 128 + b * 16 + t * 8 + lr) * 2);
                unsigned pk[2][2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { float v = (xv(b, j, t)[e] - mr[0]) * mr[1]; v = v * gw[j][e]; o[e] = v + gb[j][e]; }
                    pk[j][0] = __builtin_bit_cast(unsigned, round_pair<T>(o[0], o[1])); pk[j][1] = __builtin_bit_cast(unsigned, round_pair<T>(o[2], o[3]));
                }
                // give away the half this lane does not store, take the neighbour's piece of the half it does
                const unsigned g0 = odd ? pk[0][0] : pk[1][0], g1 = odd ? pk[0][1] : pk[1][1];
                const unsigned r0 = (unsigned)__builtin_amdgcn_mov_dpp((int)g0, 0xB1, 0xf, 0xf, true), r1 = (unsigned)__builtin_amdgcn_mov_dpp((int)g1, 0xB1, 0xf, 0xf, true);      // quad_perm [1, 0, 3, 2]
                const u32x4 d = odd ? u32x4{r0, r1, pk[1][0], pk[1][1]} : u32x4{pk[0][0], pk[0][1], r0, r1};
                __builtin_amdgcn_raw_buffer_store_b128(d, ru, uoff - lk * 8 + ucol, usoff + (b * 2 + t) * usoff8, 0);
                __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 1" ::: "memory"); __builtin_amdgcn_sched_barrier(0);       // pp_store_b128's wait states
            }
    }
    store_x();
    lds_barrier();         // the statistics area (and `fail`) is re-used by the next tile, and the ring slot is re-staged soon after
    asm volatile("s_waitcnt vmcnt(32)" ::: "memory");      // the 16 normalised-row stores are older and have landed; at most the 32 X stores stay in flight
    return true;
#undef xv
}

// FLAGS: 0 in the product.  Ablation builds exist only under -DVITX_LAB (tools/gemm_lab): 1 = no s_setprio around the MFMAs,
// 4 = no LDS-DMA in the loop, 8 = no fragment reads, 16 = no MFMAs, 32 = s_memtime stamp after every barrier of K-tiles 4..7 of the first
// tile (written to g.pos as [block][wave][64] u32), 512 = direct (unstaged) epilogue, 2048 = no epilogue at all.
// LNF: EPI_BIAS_RESID with the LayerNorm of the output rows computed in the epilogue (GemmLn, kernels.h; pp_epilogue_ln below).
template <typename T, int EPI, int FLAGS, bool LNF = false>
__global__ __launch_bounds__(512, 2) __attribute__((amdgpu_num_vgpr(PP_MAX_VGPR))) void gemm_pp_kernel(GemmArgs g, GemmLn ln) {
    using namespace pp;
    typedef typename Elem<T>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;            // wave row (= ping-pong group) / wave column

    // ---- tile walk: virtual id v keeps v % 8 == bid % 8 (same XCD), then the XCD-contiguous GROUP_M raster.
    // LNF: an XCD owns whole ROW BLOCKS and walks them column-fastest, and the launcher makes the workgroups per XCD a multiple of
    // the column tiles: the ntn tiles of a row block then always run in the SAME round on ntn neighbouring workgroups of one XCD --
    // the peers the LayerNorm epilogue exchanges row statistics with.
    const int ntm = g.M / BM, ntn = g.N_pad / BN, ntiles = ntm * ntn;
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7;
    int my_tiles_, lid_base_;
    if constexpr (LNF) {
        const int qb = ntm >> 3, rb = ntm & 7;
        const int rows_x = qb + (xcd < rb ? 1 : 0), wgx = nblk >> 3, j = bid >> 3;
        lid_base_ = (xcd * qb + min(xcd, rb)) * ntn;                   // first tile of this XCD in (row block, column tile) order
        my_tiles_ = j < rows_x * ntn ? (rows_x * ntn - j + wgx - 1) / wgx : 0;
    } else {
        const int q8 = ntiles >> 3, r8 = ntiles & 7;
        lid_base_ = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8);
        my_tiles_ = (ntiles - bid + nblk - 1) / nblk;
    }
    const int my_tiles = my_tiles_, lid_base = lid_base_;
    const int group_m = g.group_m > 0 ? g.group_m : GROUP_M;
    auto tile_origin = [&](int round, int &m0, int &n0) {
        const int v = bid + round * nblk;
        const int lid = lid_base + (v >> 3);
        if constexpr (LNF) {
            const int mb = lid / ntn;
            m0 = mb * BM; n0 = (lid - mb * ntn) * BN;
        } else {
            const int per_group = group_m * ntn;
            const int grp = lid / per_group, within = lid - grp * per_group;
            const int gm = min(group_m, ntm - grp * group_m);
            const int tn = within / gm;
            m0 = (grp * group_m + (within - tn * gm)) * BM; n0 = tn * BN;
        }
    };
    if (my_tiles <= 0) return;
#ifdef VITX_LAB
    const long long lab_c0 = __builtin_readcyclecounter(), lab_r0 = __builtin_amdgcn_s_memrealtime();      // laboratory clock probe (FLAGS 4096)
#endif

    // ---- consumer side of a LayerNorm-fusing GEMM (GemmArgs::fix): A = that GEMM's normalised rows.  Row blocks it left to the fix-up
    // (a peer workgroup did not answer in time) are recomputed from X HERE, by every workgroup for the row blocks of ITS OWN tiles, before
    // anything of A is loaded: no launch in between, no waiting on another workgroup (several redo a block: the same bits).
    if constexpr (!LNF) {
        if (ln.todo) {
            bool any = false;
            for (int r = 0; r < my_tiles; ++r) {
                int m0, n0; tile_origin(r, m0, n0);
                if (__builtin_nontemporal_load(ln.todo + m0 / BM) != ln.epoch) continue;        // workgroup-uniform
                any = true;
                const int D = g.K;
                for (int row = m0 + wave; row < m0 + BM; row += 8) {
                    const float *xr = ln.x + (size_t)row * D; T *yr = (T *)ln.out + (size_t)row * D;
                    switch (D >> 8) {
                    case 1: ln_row_tiled<T, 1>(xr, ln.w, ln.b, yr, ln.eps, lane); break;
                    case 2: ln_row_tiled<T, 2>(xr, ln.w, ln.b, yr, ln.eps, lane); break;
                    case 3: ln_row_tiled<T, 3>(xr, ln.w, ln.b, yr, ln.eps, lane); break;
                    default: ln_row_tiled<T, 4>(xr, ln.w, ln.b, yr, ln.eps, lane); break;
                    }
                }
            }
            if (any) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
        }
    }

    // ---- LDS-DMA: physical 16-B piece p = i*512 + tid of a half-tile image <-> logical (image row, slot)
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
    // (global_load_lds with 64-bit per-lane addresses made the stage issue 2.5x slower: profiles/r02_gemm_pp_lab.txt)
    __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc((void *)g.A, 0, (int)0xffffffffu, 0x00020000);
    __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc((void *)g.W, 0, (int)0xffffffffu, 0x00020000);
    __amdgpu_buffer_rsrc_t rsrcO = __builtin_amdgcn_make_buffer_rsrc(g.out, 0, (int)0xffffffffu, 0x00020000);
    auto stage_a = [&](int h, int lds_off) {
        char *base = smem + lds_off + wave * 1024;
        const int so = (is_a + is_kt * BK + h * a_half) * 2;
#pragma unroll
        for (int i = 0; i < STAGE_OPS; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, LPTR(base + i * 8192), 16, aoff[i] * 2, so, 0, LNF ? PP_AUX_A_LNF : PP_AUX_A);
    };
    auto stage_w = [&](int h, int lds_off) {
        char *base = smem + lds_off + wave * 1024;
        const int so = (is_w + is_kt * BK + h * w_half) * 2;
#pragma unroll
        for (int i = 0; i < STAGE_OPS; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, LPTR(base + i * 8192), 16, woff[i] * 2, so, 0, 0);
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

    // ---- fragment read addresses (bytes within a buffer).  A 16-row fragment of v_mfma_f32_16x16x32 is lane & 15 = row,
    // lane >> 4 = which 8 of the 32 k values of a k-step; A rows wr*64 + .. of the half-tile image, B rows wc*32 + ..
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
    f32x4 acc16[8][4];
    auto read_a = [&](int buf, int h) {      // tile t (16 rows) of the half, k-step k2 -> fa[t >> 1][(t & 1) * 2 + k2]
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) fa[t >> 1][(t & 1) * 2 + k2] = *(const v8 *)(smem + off_a(buf, h) + (t >> 1) * 4096 + rdA16[t & 1][k2]);
    };
    auto read_b = [&](int buf, int h) {      // tile u (16 columns) of the half, k-step k2 -> fb[h][u * 2 + k2]
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) fb[h][u * 2 + k2] = *(const v8 *)(smem + off_b(buf, h) + rdB16[u][k2]);
    };
    // one C quadrant: 16 MFMAs of 16 cycles; k-step major, 8 independent accumulators in between
    auto mma = [&](int ha, int hb) {
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    acc16[ha * 4 + t][hb * 2 + u] = Elem<T>::mfma16(fb[hb][u * 2 + k2], fa[t >> 1][(t & 1) * 2 + k2], acc16[ha * 4 + t][hb * 2 + u]);
    };

    unsigned stamps = 0; int n_stamp = -1;          // timeline experiment (FLAGS 32): lane i of `stamps` = i-th stamp
    auto stamp = [&]() {
        if constexpr ((FLAGS & 32) != 0) {
            if (n_stamp >= 0 && n_stamp < 64) {
                const unsigned t = (unsigned)__builtin_readcyclecounter();
                stamps = lane == n_stamp ? t : stamps;
                ++n_stamp;
            }
        }
    };
    // one phase = [reads, stage] | counted wait | barrier | 16 MFMAs | barrier
#define PP_PHASE(READS, STAGE, VMCNT, HA, HB, FIRST, FIRST_STMT)                                              \
    {                                                                                      \
        if (!(FLAGS & 8)) { READS; }                                                       \
        if (FIRST) { FIRST_STMT; }                                                         \
        if (!(FLAGS & 4)) { STAGE; }                                                       \
        if (FIRST) { if (relaxed) pp_wait_vmcnt<VMCNT + 1 + pp_epi_stores<EPI>()>(); else pp_wait_vmcnt<VMCNT + 1>(); }   \
        else pp_wait_vmcnt<VMCNT>();                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        pp_barrier();                                                                      \
        stamp();                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        if (!(FLAGS & 1)) __builtin_amdgcn_s_setprio(1);                                   \
        if (!(FLAGS & 16)) mma(HA, HB);                                                    \
        if (!(FLAGS & 1)) __builtin_amdgcn_s_setprio(0);                                   \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        pp_barrier();                                                                      \
        stamp();                                                                           \
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
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) acc16[ha * 4 + t][hb * 2 + u] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
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
    typedef std::integral_constant<int, 0> I0; typedef std::integral_constant<int, 1> I1;

    // ---- prologue: A0 B0 B1 A1 of K-tile 0 and A0 B0 of K-tile 1 in flight, the first two landed
    stage_a(0, off_a(0, 0)); stage_w(0, off_b(0, 0)); stage_w(1, off_b(0, 1)); stage_a(1, off_a(0, 1)); advance();     // nkt >= 2: K-tile 1 exists
    stage_a(0, off_a(1, 0)); stage_w(0, off_b(1, 0));
    pp_wait_vmcnt<4 * STAGE_OPS>();
    pp_barrier();
    if (wr == 1) pp_barrier();                      // the second wave row runs one barrier behind the first

    for (int round = 0; round < my_tiles; ++round) {
        int m0, n0; tile_origin(round, m0, n0);
        bias_so = __builtin_amdgcn_readfirstlane((n0 + wc * 64) * 4);
        for (int kt = 0; kt < nkt; kt += 2) {
            if constexpr ((FLAGS & 32) != 0) { if (round == 0 && kt == 4) n_stamp = 0; }
            ktile(I0{}, kt == 0); ktile(I1{}, false);
        }
        if constexpr ((FLAGS & 8) != 0) {           // fragments never read: keep the MFMA operands "defined" for the compiler
            if (round == 0) { for (int ii = 0; ii < 2; ++ii) for (int ks = 0; ks < 4; ++ks) { asm volatile("" : "+v"(fa[ii][ks])); asm volatile("" : "+v"(fb[ii][ks])); } }
        }
        if constexpr ((FLAGS & 16) != 0) { for (int ii = 0; ii < 2; ++ii) for (int ks = 0; ks < 4; ++ks) { asm volatile("" :: "v"(fa[ii][ks]), "v"(fb[ii][ks])); } }
        const bool full = LNF || ((m0 + BM <= g.M_real) && (n0 + BN <= g.N));       // LNF: the padded rows are computed and stored too (GemmLn)
        // The second wave row runs one barrier behind, so its last K-loop barrier would only be released by the first row's first
        // barrier of the NEXT tile -- i.e. after the first row's epilogue, and the two rows' epilogues would run one after the other.
        // Aligning the rows here (and restoring the offset after the epilogue) lets both epilogues run at the same time: forward
        // 10.82 -> 10.75 ms, fc1 +4 % (r02c).
        if (wr == 0) pp_barrier();
        if constexpr ((FLAGS & 2048) != 0) {
#pragma unroll
            for (int t = 0; t < 8; ++t) asm volatile("" :: "v"(acc16[t][0]), "v"(acc16[t][1]), "v"(acc16[t][2]), "v"(acc16[t][3]));
        } else if (full && EPI != EPI_PATCH && !(FLAGS & 512)) {
            constexpr int esz = (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_HILO) ? 2 : 4;
            // row layout of the stores: lane -> row (lane>>3) + 8t of a 32-row block, 16-byte piece lane&7 of the wave's 128-byte row segment
            const int voff = ((wr * 128 + (lane >> 3)) * g.ldo + wc * 64) * esz + (lane & 7) * 16;
            f32x4 bq[4];                       // bias of columns u * 16 + 4 g4 .. + 3, staged into the wave's patch by LDS-DMA during the K loop
#pragma unroll
            for (int u = 0; u < 4; ++u) bq[u] = *(const f32x4 *)(smem + LDS + wave * 4096 + u * 64 + g4 * 16);
            // readfirstlane: the tile origin comes out of an integer division done on the VALU; without it hipcc wraps every
            // buffer op in a waterfall loop over the (uniform) SGPR offset
            if constexpr (LNF) relaxed = pp_epilogue_ln<T>(g, ln, acc16, rsrcO, smem, voff, __builtin_amdgcn_readfirstlane((m0 * g.ldo + n0) * esz), 8 * g.ldo * esz, tid, m0, n0, ntn);
            else {
                epilogue16_staged<T, EPI, 4, (EPI == EPI_BIAS_GELU ? PP_AUX_H : ((EPI == EPI_BIAS || EPI == EPI_BIAS_HILO) ? PP_AUX_QKV : 0))>(acc16, bq, rsrcO, smem + LDS + wave * 4096, voff, __builtin_amdgcn_readfirstlane((m0 * g.ldo + n0) * esz), 8 * g.ldo * esz, lane, (int)(g.hilo_off * esz));
                relaxed = true;
            }
        } else {
            const int row0 = m0 + wr * 128 + l15, ncol = n0 + wc * 64;
            if (full) epilogue16<T, EPI, 8, 4, true>(g, acc16, row0, ncol + 4 * g4);
            else epilogue16<T, EPI, 8, 4, false>(g, acc16, row0, ncol + 4 * g4);
        }
        if (wr == 1) pp_barrier();
    }
    if (wr == 0) pp_barrier();
    pp_wait_vmcnt<0>();                             // the trailing (unused) stages must land before the LDS allocation is released
    if constexpr ((FLAGS & 32) != 0) ((unsigned *)g.pos)[((size_t)bid * 8 + wave) * 64 + lane] = stamps;
#ifdef VITX_LAB
    if constexpr ((FLAGS & 4096) != 0) {
        if (g.pos && tid == 0) {
            long long *o = (long long *)g.pos + (size_t)bid * 4;
            o[0] = __builtin_readcyclecounter() - lab_c0; o[1] = __builtin_amdgcn_s_memrealtime() - lab_r0; o[2] = my_tiles; o[3] = g.K / BK;
        }
    }
#endif
#undef PP_PHASE
}

bool gemm_pp_supports(const GemmArgs &a) {
    // byte offsets into A, W and out are 32-bit (buffer addressing)
    const size_t lim = 0xf0000000u;
    if ((size_t)a.M * a.lda * 2 > lim || (size_t)a.N_pad * a.ldw * 2 > lim || (size_t)(a.M + a.M / 64 + 2) * a.ldo * 4 + (size_t)(a.hilo_off > 0 ? a.hilo_off : 0) * 2 > lim) return false;
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
    hipLaunchKernelGGL((gemm_pp_kernel<T, EPI, FLAGS>), dim3(grid), dim3(512), pp::LDS_ALL, stream, a, a.fix ? *a.fix : GemmLn{});
    return hipGetLastError();
}
// Persistent grid of the LayerNorm-fusing GEMM: 8 XCDs x wgx workgroups, wgx a multiple of the ntn column tiles (so the tiles of a row
// block share a round) and balanced over the rounds the busiest XCD needs.
int gemm_pp_ln_grid(int n_cu, int M, int N) {
    const int ntm = M / pp::BM, ntn = N / pp::BN;
    int per_xcd = (n_cu > 0 ? n_cu : 256) / 8;
    if (per_xcd < ntn) per_xcd = ntn;
    const int cap = per_xcd / ntn * ntn;
    const int most = ((ntm + 7) / 8) * ntn;                         // tiles of the XCD with the most row blocks
    const int rounds = (most + cap - 1) / cap;
    int wgx = ((most + rounds - 1) / rounds + ntn - 1) / ntn * ntn;
    if (wgx > cap) wgx = cap;
    return 8 * wgx;
}
template <typename T>
static hipError_t launch_pp_ln(const GemmArgs &a, int n_cu, hipStream_t stream, bool prepare) {
    if (prepare) return hipFuncSetAttribute((const void *)gemm_pp_kernel<T, EPI_BIAS_RESID, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, pp::LDS_ALL);
    const GemmLn &ln = *a.ln;
    if (!ln.w || !ln.b || !ln.out || !ln.sync || !ln.todo || !ln.fallbacks || !ln.epoch || a.N != a.N_pad || a.N != a.ldo || a.N / pp::BN > LN_MAX_TILES) return hipErrorInvalidValue;
    hipLaunchKernelGGL((gemm_pp_kernel<T, EPI_BIAS_RESID, 0, true>), dim3(gemm_pp_ln_grid(n_cu, a.M, a.N)), dim3(512), pp::LDS_ALL, stream, a, ln);
    return hipGetLastError();
}
template <typename T>
static hipError_t launch_pp_t(int epi, const GemmArgs &a, int n_cu, hipStream_t stream, int flags, bool prepare) {
#ifdef VITX_LAB
    if (flags) {       // ablation builds (tools/gemm_lab) exist for the plain bias epilogue only
        if (epi != EPI_BIAS) return hipErrorInvalidValue;
        switch (flags) {
        case 1: return launch_pp_inst<T, EPI_BIAS, 1>(a, n_cu, stream, prepare);
        case 4: return launch_pp_inst<T, EPI_BIAS, 4>(a, n_cu, stream, prepare);
        case 8: return launch_pp_inst<T, EPI_BIAS, 8>(a, n_cu, stream, prepare);
        case 12: return launch_pp_inst<T, EPI_BIAS, 12>(a, n_cu, stream, prepare);
        case 16: return launch_pp_inst<T, EPI_BIAS, 16>(a, n_cu, stream, prepare);
        case 32: return launch_pp_inst<T, EPI_BIAS, 32>(a, n_cu, stream, prepare);
        case 512: return launch_pp_inst<T, EPI_BIAS, 512>(a, n_cu, stream, prepare);
        case 2048: return launch_pp_inst<T, EPI_BIAS, 2048>(a, n_cu, stream, prepare);
        case 4096: return launch_pp_inst<T, EPI_BIAS, 4096>(a, n_cu, stream, prepare);
        default: return hipErrorInvalidValue;
        }
    }
#else
    if (flags) return hipErrorInvalidValue;
#endif
    if (a.ln || (prepare && epi == EPI_BIAS_RESID)) {      // bring-up prepares both builds of the residual epilogue
        if (epi != EPI_BIAS_RESID) return hipErrorInvalidValue;
        const hipError_t e = launch_pp_ln<T>(a, n_cu, stream, prepare);
        if (!prepare || e != hipSuccess) return e;
    }
    switch (epi) {
    case EPI_BIAS: return launch_pp_inst<T, EPI_BIAS, 0>(a, n_cu, stream, prepare);
    case EPI_BIAS_GELU: return launch_pp_inst<T, EPI_BIAS_GELU, 0>(a, n_cu, stream, prepare);
    case EPI_BIAS_RESID: return launch_pp_inst<T, EPI_BIAS_RESID, 0>(a, n_cu, stream, prepare);
    case EPI_BIAS_F32: return launch_pp_inst<T, EPI_BIAS_F32, 0>(a, n_cu, stream, prepare);
    case EPI_PATCH: return launch_pp_inst<T, EPI_PATCH, 0>(a, n_cu, stream, prepare);
    case EPI_BIAS_HILO: return launch_pp_inst<T, EPI_BIAS_HILO, 0>(a, n_cu, stream, prepare);
    default: return hipErrorInvalidValue;
    }
}
hipError_t launch_gemm_pp(int dtype, int epi, const GemmArgs &a, int n_cu, hipStream_t stream, int flags, bool prepare) {
    if (!prepare && !gemm_pp_supports(a)) return hipErrorInvalidValue;
    return dtype == DT_F16 ? launch_pp_t<_Float16>(epi, a, n_cu, stream, flags, prepare) : launch_pp_t<__bf16>(epi, a, n_cu, stream, flags, prepare);
}

}  // namespace vitx

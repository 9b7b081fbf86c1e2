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

// whole-row stores of the staged epilogue per wave and tile (epilogue16_staged): the next tile's K loop skips over exactly this many
template <int EPI> __host__ __device__ constexpr int pp_epi_stores() { return (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) ? 16 : 32; }

// FLAGS: 0 in the product.  Ablation builds exist only under -DVITX_LAB (tools/gemm_lab): 1 = no s_setprio around the MFMAs,
// 4 = no LDS-DMA in the loop, 8 = no fragment reads, 16 = no MFMAs, 32 = s_memtime stamp after every barrier of K-tiles 4..7 of the first
// tile (written to g.pos as [block][wave][64] u32), 512 = direct (unstaged) epilogue, 2048 = no epilogue at all.
template <typename T, int EPI, int FLAGS>
__global__ __launch_bounds__(512, 2) __attribute__((amdgpu_num_vgpr(PP_MAX_VGPR))) void gemm_pp_kernel(GemmArgs g) {
    using namespace pp;
    typedef typename Elem<T>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
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
        for (int i = 0; i < STAGE_OPS; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, LPTR(base + i * 8192), 16, aoff[i] * 2, so, 0, 0);
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
        const bool full = (m0 + BM <= g.M_real) && (n0 + BN <= g.N);
        // The second wave row runs one barrier behind, so its last K-loop barrier would only be released by the first row's first
        // barrier of the NEXT tile -- i.e. after the first row's epilogue, and the two rows' epilogues would run one after the other.
        // Aligning the rows here (and restoring the offset after the epilogue) lets both epilogues run at the same time: forward
        // 10.82 -> 10.75 ms, fc1 +4 % (r02c).
        if (wr == 0) pp_barrier();
        if constexpr ((FLAGS & 2048) != 0) {
#pragma unroll
            for (int t = 0; t < 8; ++t) asm volatile("" :: "v"(acc16[t][0]), "v"(acc16[t][1]), "v"(acc16[t][2]), "v"(acc16[t][3]));
        } else if (full && EPI != EPI_PATCH && !(FLAGS & 512)) {
            constexpr int esz = (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) ? 2 : 4;
            // row layout of the stores: lane -> row (lane>>3) + 8t of a 32-row block, 16-byte piece lane&7 of the wave's 128-byte row segment
            const int voff = ((wr * 128 + (lane >> 3)) * g.ldo + wc * 64) * esz + (lane & 7) * 16;
            f32x4 bq[4];                       // bias of columns u * 16 + 4 g4 .. + 3, staged into the wave's patch by LDS-DMA during the K loop
#pragma unroll
            for (int u = 0; u < 4; ++u) bq[u] = *(const f32x4 *)(smem + LDS + wave * 4096 + u * 64 + g4 * 16);
            // readfirstlane: the tile origin comes out of an integer division done on the VALU; without it hipcc wraps every
            // buffer op in a waterfall loop over the (uniform) SGPR offset
            epilogue16_staged<T, EPI, 4>(acc16, bq, rsrcO, smem + LDS + wave * 4096, voff, __builtin_amdgcn_readfirstlane((m0 * g.ldo + n0) * esz), 8 * g.ldo * esz, lane);
            relaxed = true;
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
#undef PP_PHASE
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
        default: return hipErrorInvalidValue;
        }
    }
#else
    if (flags) return hipErrorInvalidValue;
#endif
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

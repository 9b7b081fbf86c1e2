// gemm_pers.hip -- persistent, stream-fed MFMA GEMM for gfx950 (the production GEMM of the forward).
//
//   C[M][N] = A[M][K] . W[N][K]^T   (+ fused epilogue), A/W fp16 or bf16, f32 accumulate.
//
// Why persistent: on the ViT shapes (K = 768) one 256x256 tile is only ~10 us of MFMA work, and the
// per-workgroup launch + cold prologue of a one-tile-per-workgroup kernel costs almost as much
// (measured 7 us, profiles/r01_gemm_ablation.txt).  Here ONE workgroup per CU walks its tiles and the
// LDS-DMA ring never drains: the K-slots of tile t+1 are already in flight while tile t's epilogue runs.
//
//   * 512 threads = 8 waves as 2(M) x 4(N); tile 256x256; each wave owns 128 x 64 of C as 4 x 2
//     MFMA 32x32x16 accumulators (128 f32/lane).  Operands are SWAPPED in the MFMA (W fragment as A,
//     activation fragment as B) so a lane holds 4 consecutive output COLUMNS per accumulator quad:
//     the epilogue stores 8-byte (fp16x4) / 16-byte (f32x4) vectors instead of scalars.
//   * K is consumed in slots of 32; LDS ring of NS slots [A: 256 x 32 | W: 256 x 32] (32 KiB each),
//     filled by global_load_lds dwordx4 NS-1 slots ahead of the MFMAs, across tile boundaries, retired
//     with a COUNTED s_waitcnt vmcnt + one raw s_barrier per slot.
//   * the loop body is MFMA-first: the 16 MFMAs of a slot are issued from fragments that are already
//     in registers; the 4 DMA issues and the 12 ds_read_b128 of the NEXT k-step are interleaved
//     between them (sched_group_barrier), so address/issue cost hides under matrix-pipe time.
//   * 64-byte LDS rows, 16-B slots XOR-swizzled per 256-B bank line (conflict-free ds_read_b128);
//     the DMA image is lane-linear, the permutation sits on the per-lane global source address.
//   * tiles are walked in XCD-contiguous GROUP_M x n raster order so co-resident tiles of one XCD
//     share A / W panels in its 4 MiB L2.
#include <stdlib.h>

#include "device_common.h"
#include "kernels.h"

namespace vitx {

namespace {

constexpr int PBM = 256, PBN = 256, PBK = 32;
constexpr int P_A_BYTES = PBM * 64, P_SLOT_BYTES = (PBM + PBN) * 64;     // 32 KiB per slot
constexpr int P_GROUP_M = 8;
constexpr int P_G = 4;               // DMA instructions per thread per slot (2 A + 2 W)

__device__ __forceinline__ int pswz_byte(int row, int s /*0..3*/) {
    const int line = row >> 2;
    const int s16 = ((row & 3) << 2) | s;
    return line * 256 + ((s16 ^ (line & 15)) << 4);
}
__device__ __forceinline__ void pswz_inv(int p, int &row, int &s) {
    const int line = p >> 4;
    const int s16 = (p & 15) ^ (line & 15);
    row = line * 4 + (s16 >> 2);
    s = s16 & 3;
}
template <int N> __device__ __forceinline__ void pwait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void pbarrier() { asm volatile("s_barrier" ::: "memory"); }

// tile id in raster order -> (m0, n0)
__device__ __forceinline__ void tile_origin(int lid, int ntm, int ntn, int &m0, int &n0) {
    const int per_group = P_GROUP_M * ntn;
    const int grp = lid / per_group, within = lid - grp * per_group;
    const int gm = min(P_GROUP_M, ntm - grp * P_GROUP_M);
    const int tn = within / gm;
    m0 = (grp * P_GROUP_M + (within - tn * gm)) * PBM;
    n0 = tn * PBN;
}

// Epilogue of one wave's 128 x 64 block.  Accumulator (i, j) holds C^T: register r of lane (l31, hh)
// is row m = i*32 + l31, column n = j*32 + 8*(r>>2) + 4*hh + (r&3): quads of 4 consecutive columns.
template <typename T, int EPI, bool FULL>
__device__ __forceinline__ void pers_epilogue(const GemmArgs &g, f32x16 (&acc)[4][2], int row0 /*+l31*/, int col0 /*+4*hh*/) {
    typedef typename Elem<T>::v4 v4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = row0 + i * 32;
        const bool row_ok = FULL || row < g.M_real;
        size_t obase;
        const float *prow = nullptr;
        if constexpr (EPI == EPI_PATCH) {
            const int b = row / g.tpi, t = row - b * g.tpi;
            obase = ((size_t)row + b + 1) * g.ldo;
            prow = g.pos + (size_t)(t + 1) * g.ldo;
        } else {
            obase = (size_t)row * g.ldo;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = col0 + j * 32 + q * 8;
                if (!FULL && !(row_ok && col < g.N)) continue;      // N is a multiple of 4: a quad is all-in or all-out
                const float4 bv = *(const float4 *)(g.bias + col);
                float v[4] = {acc[i][j][q * 4 + 0] + bv.x, acc[i][j][q * 4 + 1] + bv.y, acc[i][j][q * 4 + 2] + bv.z, acc[i][j][q * 4 + 3] + bv.w};
                if constexpr (EPI == EPI_BIAS) {
                    *(v4 *)((T *)g.out + obase + col) = v4{(T)v[0], (T)v[1], (T)v[2], (T)v[3]};
                } else if constexpr (EPI == EPI_BIAS_GELU) {
                    *(v4 *)((T *)g.out + obase + col) = v4{(T)gelu_tanh(rnd<T>(v[0])), (T)gelu_tanh(rnd<T>(v[1])), (T)gelu_tanh(rnd<T>(v[2])), (T)gelu_tanh(rnd<T>(v[3]))};
                } else if constexpr (EPI == EPI_BIAS_RESID || EPI == EPI_BIAS_F32) {   // RESID: the residual was the accumulator's initial value
                    *(float4 *)((float *)g.out + obase + col) = float4{v[0], v[1], v[2], v[3]};
                } else {   // EPI_PATCH
                    const float4 pv = *(const float4 *)(prow + col);
                    *(float4 *)((float *)g.out + obase + col) = float4{v[0] + pv.x, v[1] + pv.y, v[2] + pv.z, v[3] + pv.w};
                }
            }
        }
    }
}

// Accumulator initialisation of one wave's block: zero, or (EPI_BIAS_RESID) the f32 residual tile itself, so
// the read half of the read-modify-write is issued a whole tile ahead of its use instead of in the epilogue.
template <int EPI>
__device__ __forceinline__ void pers_acc_init(const GemmArgs &g, f32x16 (&acc)[4][2], int row0, int col0, bool full, bool valid) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 v = float4{0, 0, 0, 0};
                if constexpr (EPI == EPI_BIAS_RESID) {
                    const int row = row0 + i * 32, col = col0 + j * 32 + q * 8;
                    if (valid && (full || (row < g.M_real && col < g.N))) v = *(const float4 *)((const float *)g.out + (size_t)row * g.ldo + col);
                }
                acc[i][j][q * 4 + 0] = v.x; acc[i][j][q * 4 + 1] = v.y; acc[i][j][q * 4 + 2] = v.z; acc[i][j][q * 4 + 3] = v.w;
            }
}

template <typename T, int EPI, int NS, bool DBG>
__global__ __launch_bounds__(512, 2) void gemm_pers_kernel(GemmArgs g) {
    const int dbg = DBG ? g.dbg : 0;     // ablation bits for experiments: 1 no DMA, 2 no ds_read, 4 no MFMA, 16 no wait/barrier
    typedef typename Elem<T>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    // ---- this workgroup's tile sequence: round r -> raster id r*nwg + xmap(bid)
    const int nwg = gridDim.x, bid = blockIdx.x;                 // nwg is a multiple of 8
    const int xmap = (bid & 7) * (nwg >> 3) + (bid >> 3);        // XCD x owns the contiguous ids [x*nwg/8, (x+1)*nwg/8) of each round
    const int ntm = g.M / PBM, ntn = g.N_pad / PBN, ntiles = ntm * ntn;
    const int my_tiles = (ntiles - xmap + nwg - 1) / nwg;        // xmap < nwg <= ntiles rounded: may be 0
    if (my_tiles <= 0) return;
    const int nslots = g.K / PBK;
    const int total = my_tiles * nslots;                         // length of this workgroup's slot stream

    // ---- DMA side: per-thread piece offsets (tile independent) + per-tile scalar bases
    const T *A = (const T *)g.A, *W = (const T *)g.W;
    int aoff[2], woff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int row, s; pswz_inv(i * 512 + tid, row, s);
        aoff[i] = row * g.lda + s * 8;
        woff[i] = row * g.ldw + s * 8;
        if (DBG && (dbg & 8)) { aoff[i] = woff[i] = (i * 512 + tid) * 8; }     // bandwidth probe: 1 KiB contiguous per wave instruction
    }
    int is_tile = 0, is_k = 0;                                   // issue cursor
    int im0, in0;
    tile_origin(xmap, ntm, ntn, im0, in0);
    const T *Ab = A + (size_t)im0 * g.lda, *Wb = W + (size_t)in0 * g.ldw;
    auto issue_dma = [&](int pos) {
        char *base = smem + pos * P_SLOT_BYTES + wave * 1024;
        const int k0 = (DBG && (dbg & 8)) ? is_k * 4096 : is_k * PBK;
#pragma unroll
        for (int i = 0; i < 2; ++i) __builtin_amdgcn_global_load_lds(GPTR(Ab + aoff[i] + k0), LPTR(base + i * 8192), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i) __builtin_amdgcn_global_load_lds(GPTR(Wb + woff[i] + k0), LPTR(base + P_A_BYTES + i * 8192), 16, 0, 0);
    };
    auto advance_cursor = [&]() {
        // past the end of the stream the cursor parks on the last valid slot (harmless re-load, never consumed)
        if (++is_k == nslots) {
            if (is_tile + 1 < my_tiles) {
                is_k = 0; ++is_tile;
                tile_origin(is_tile * nwg + xmap, ntm, ntn, im0, in0);
                Ab = A + (size_t)im0 * g.lda; Wb = W + (size_t)in0 * g.ldw;
            } else {
                is_k = nslots - 1;
            }
        }
    };

    // ---- fragment read offsets within a slot
    int a_rd[4][2], w_rd[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a_rd[i][ks] = pswz_byte(wm * 128 + i * 32 + l31, ks * 2 + hh);
#pragma unroll
        for (int j = 0; j < 2; ++j) w_rd[j][ks] = P_A_BYTES + pswz_byte(wn * 64 + j * 32 + l31, ks * 2 + hh);
    }

    int pos = 0, ck = 0, ctile = 0;            // compute cursor: ring position, k-slot in tile, tile index
    int cm0, cn0;
    tile_origin(xmap, ntm, ntn, cm0, cn0);
    f32x16 acc[4][2];
    pers_acc_init<EPI>(g, acc, cm0 + wm * 128 + l31, cn0 + wn * 64 + 4 * hh, (cm0 + PBM <= g.M_real) && (cn0 + PBN <= g.N), true);

    // ---- prologue: NS-1 slots in flight; slots 0 and 1 landed
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) { issue_dma(s); advance_cursor(); }
    pwait_vmcnt<(NS - 3) * P_G>();
    pbarrier();

    v8 fa[2][4], fw[2][2];
    auto load_frags = [&](int buf, int pos, int ks) {
        const char *sb = smem + pos * P_SLOT_BYTES;
#pragma unroll
        for (int j = 0; j < 2; ++j) fw[buf][j] = *(const v8 *)(sb + w_rd[j][ks]);
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[buf][i] = *(const v8 *)(sb + a_rd[i][ks]);
    };
    auto mma = [&](int buf) {   // swapped operands: result tile is C^T (rows = n, cols = m)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = Elem<T>::mfma(fw[buf][j], fa[buf][i], acc[i][j]);
    };
    auto tile_end = [&]() {     // fused epilogue, then re-arm the accumulators for the next tile of the stream
        const bool full = (cm0 + PBM <= g.M_real) && (cn0 + PBN <= g.N);
        if (full) pers_epilogue<T, EPI, true>(g, acc, cm0 + wm * 128 + l31, cn0 + wn * 64 + 4 * hh);
        else pers_epilogue<T, EPI, false>(g, acc, cm0 + wm * 128 + l31, cn0 + wn * 64 + 4 * hh);
        ck = 0; ++ctile;
        tile_origin(ctile * nwg + xmap, ntm, ntn, cm0, cn0);
        pers_acc_init<EPI>(g, acc, cm0 + wm * 128 + l31, cn0 + wn * 64 + 4 * hh, (cm0 + PBM <= g.M_real) && (cn0 + PBN <= g.N), ctile < my_tiles);
    };

    // Role split (one barrier per slot, both groups in lockstep at the barrier, complementary in between):
    //   waves 0-3 (wm = 0, one per SIMD):  DMA issue + fragment reads of slot s FIRST, its 16 MFMAs LAST;
    //   waves 4-7 (wm = 1, one per SIMD):  16 MFMAs of slot s FIRST (fragments read in the previous iteration),
    //                                      then DMA issue + fragment reads of slot s+1.
    // Each SIMD hosts one wave of each group, so its matrix pipe is fed by one group while the other group's
    // VMEM/LDS instructions issue, instead of both stalling on the same instruction mix at the same time.
    if (wm == 0) {
        for (int s = 0; s < total; ++s) {
            const int pos_next = (pos + 1 == NS) ? 0 : pos + 1;
            const int pos_fill = (pos == 0) ? NS - 1 : pos - 1;     // slot s-1's position, freed by the barrier that ended iteration s-1
            if (!(dbg & 1)) issue_dma(pos_fill);
            if (!(dbg & 2)) { load_frags(0, pos, 0); load_frags(1, pos, 1); }
            advance_cursor();
            if (!(dbg & 64)) __builtin_amdgcn_s_setprio(1);
            if (!(dbg & 4)) { mma(0); mma(1); }
            if (dbg & 32) { mma(0); mma(1); }
            __builtin_amdgcn_s_setprio(0);
            if (++ck == nslots) tile_end();
            // slot s+2 must have landed (every wave's pieces) before iteration s+1 reads it; NS-3 younger slots stay in flight
            if (!(dbg & 16)) { pwait_vmcnt<(NS - 3) * P_G>(); pbarrier(); }
            pos = pos_next;
        }
    } else {
        load_frags(0, 0, 0);
        load_frags(1, 0, 1);
        for (int s = 0; s < total; ++s) {
            const int pos_next = (pos + 1 == NS) ? 0 : pos + 1;
            const int pos_fill = (pos == 0) ? NS - 1 : pos - 1;
            if (!(dbg & 64)) __builtin_amdgcn_s_setprio(1);
            if (!(dbg & 4)) { mma(0); mma(1); }
            if (dbg & 32) { mma(0); mma(1); }
            __builtin_amdgcn_s_setprio(0);
            if (++ck == nslots) tile_end();
            if (!(dbg & 1)) issue_dma(pos_fill);
            if (!(dbg & 2)) { load_frags(0, pos_next, 0); load_frags(1, pos_next, 1); }      // slot s+1 landed at the end of iteration s-1
            advance_cursor();
            if (!(dbg & 16)) { pwait_vmcnt<(NS - 3) * P_G>(); pbarrier(); }
            pos = pos_next;
        }
    }
    pwait_vmcnt<0>();      // parked re-loads must not outlive the workgroup's LDS allocation
}

template <typename T, int EPI, int NS, bool DBG = false>
hipError_t launch_pers_inst(const GemmArgs &a, hipStream_t stream) {
    constexpr int lds = NS * P_SLOT_BYTES;
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute((const void *)gemm_pers_kernel<T, EPI, NS, DBG>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr_set = true; }
    static int n_cu = 0;
    if (!n_cu) { int dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev); if (n_cu < 8) n_cu = 8; }
    const int ntiles = (a.M / PBM) * (a.N_pad / PBN);
    int grid = std::min(ntiles, n_cu) & ~7;          // multiple of 8 (one contiguous id range per XCD)
    if (grid < 8) grid = 8;
    hipLaunchKernelGGL((gemm_pers_kernel<T, EPI, NS, DBG>), dim3(grid), dim3(512), lds, stream, a);
    return hipGetLastError();
}

template <typename T, int NS>
hipError_t launch_pers_t(const GemmArgs &a, int epi, hipStream_t stream) {
    if (a.dbg) return epi == EPI_BIAS ? launch_pers_inst<T, EPI_BIAS, NS, true>(a, stream) : hipErrorInvalidValue;
    switch (epi) {
    case EPI_BIAS: return launch_pers_inst<T, EPI_BIAS, NS>(a, stream);
    case EPI_BIAS_GELU: return launch_pers_inst<T, EPI_BIAS_GELU, NS>(a, stream);
    case EPI_BIAS_RESID: return launch_pers_inst<T, EPI_BIAS_RESID, NS>(a, stream);
    case EPI_BIAS_F32: return launch_pers_inst<T, EPI_BIAS_F32, NS>(a, stream);
    case EPI_PATCH: return launch_pers_inst<T, EPI_PATCH, NS>(a, stream);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace

bool gemm_pers_supports(const GemmArgs &a) {
    return a.M % PBM == 0 && a.N_pad % PBN == 0 && a.K % PBK == 0 && a.K >= 3 * PBK && a.N % 4 == 0 && a.ldo % 4 == 0 &&
           (long)(a.M / PBM) * (a.N_pad / PBN) >= 8;
}

// cfg 904 / 905: ring depth 4 / 5
hipError_t launch_gemm_pers(int dtype, int epi, const GemmArgs &a0, int cfg, hipStream_t stream) {
    if (!gemm_pers_supports(a0)) return hipErrorInvalidValue;
    static int dbg = -1;
    if (dbg < 0) { const char *e = getenv("VITX_GEMM_DBG"); dbg = e ? atoi(e) : 0; }
    GemmArgs a = a0; a.dbg = dbg;
    if (cfg == 904) return dtype == DT_F16 ? launch_pers_t<_Float16, 4>(a, epi, stream) : launch_pers_t<__bf16, 4>(a, epi, stream);
    return dtype == DT_F16 ? launch_pers_t<_Float16, 5>(a, epi, stream) : launch_pers_t<__bf16, 5>(a, epi, stream);
}

}  // namespace vitx

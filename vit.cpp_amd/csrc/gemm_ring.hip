// gemm_ring.hip -- the main MFMA GEMM of the ViT forward path for gfx950.
//
//   C[M][N] = A[M][K] . W[N][K]^T   (+ fused epilogue), A/W fp16 or bf16, f32 accumulate.
//
// Design (MI355X-first, see DESIGN.md "GEMM"):
//   * waves are laid out 2(M) x 4(N); each wave owns (WMT*32) x 64 of C as WMT x 2 MFMA 32x32x16
//     accumulators: 8 waves, tile 256x256 (WMT=4, cfg 445/945) or 128x256 (WMT=2, cfg 245: short M and
//     the remainder rows of a launch), one workgroup per CU, ring of 5 slots (160 / 120 KiB LDS).
//   * cfg 945 (default for the wide tile) is PERSISTENT: one workgroup per CU walks its tiles and keeps
//     the ring running across tile boundaries (gemm_stream_kernel below).
//   * K is consumed in slots of 32: an LDS ring of NS slots, each [A: BM x 32 | W: BN x 32] halves.
//     Slots are filled by LDS-DMA (global_load_lds dwordx4) issued NS-1 slots ahead and retired with
//     COUNTED s_waitcnt vmcnt(N) + one raw s_barrier per slot, so loads stay in flight across
//     barriers (the compiler's __syncthreads() would drain them).
//   * fragment registers are double buffered across 16-deep k-steps: the ds_read_b128 of step j+1
//     (possibly in the next, already-landed slot) are issued under the MFMAs of step j -- MFMA FIRST, then
//     one read per MFMA (sched_group_barrier), so the lgkmcnt(0) in front of a k-step waits on reads that
//     are 2+ MFMAs old instead of on the reads just issued (+5-7 %).
//   * 64-byte LDS rows, 4 rows per 256-B bank line, 16-B slots XOR-ed with (line & 15): every
//     ds_read_b128 lane group hits 16 distinct slots (conflict-free); the LDS image itself is
//     lane-linear (DMA requirement), the permutation is applied to the per-lane global source address.
//   * workgroup -> tile map: XCD-contiguous ids, then GROUP_M x n blocks so the tiles co-resident on
//     one XCD share A and W panels in that XCD's 4 MiB L2.
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>

#include "device_common.h"
#include "kernels.h"

#ifndef RING_SCHED
#define RING_SCHED 1
#endif
namespace vitx {

constexpr int GROUP_M = 8;          // m-tiles per raster group

__device__ __forceinline__ int swz64_byte(int row, int s /*0..3*/) {
    const int line = row >> 2;
    const int s16 = ((row & 3) << 2) | s;
    return line * 256 + ((s16 ^ (line & 15)) << 4);
}
__device__ __forceinline__ void swz64_inv(int p, int &row, int &s) {
    const int line = p >> 4;
    const int s16 = (p & 15) ^ (line & 15);
    row = line * 4 + (s16 >> 2);
    s = s16 & 3;
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wg_barrier() { asm volatile("s_barrier" ::: "memory"); }

// Bias of the wave's WNT column tiles (this lane's column of each), 0 beyond N.
template <int WNT>
__device__ __forceinline__ void load_bias(const GemmArgs &g, int col0, float (&bv)[WNT]) {
#pragma unroll
    for (int j = 0; j < WNT; ++j) { const int c = col0 + j * 32; bv[j] = c < g.N ? g.bias[c] : 0.0f; }
}

// Fused epilogue of one wave's (WMT*32) x 64 accumulator block.  FULL tiles skip every bounds check.
// C layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
template <typename T, int EPI, int WMT, int WNT, bool FULL>
__device__ __forceinline__ void epilogue(const GemmArgs &g, f32x16 (&acc)[WMT][WNT], int row0, int col0, const float (&bv)[WNT]) {
    bool col_ok[WNT];
#pragma unroll
    for (int j = 0; j < WNT; ++j) col_ok[j] = FULL || (col0 + j * 32) < g.N;
    if constexpr (EPI == EPI_BIAS_RESID) {
        // read-modify-write of the f32 residual stream in units of 16 loads (one 32x32 accumulator tile): the loads of unit
        // u+1 are issued BEFORE the adds/stores of unit u, so only the first unit's load latency is exposed.  (Two 32-load
        // batches in flight need 32 more registers than the K loop leaves: 60 spills, fc2 340 -> 474 us.)
        float res[2][16];
        auto load_unit = [&](int u, float (&dst)[16]) {          // unit u = (32-row block u/WNT, column tile u%WNT): 16 rows x this lane's column
            const int i = u / WNT, j = u % WNT;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + i * 32 + (r & 3) + 8 * (r >> 2);
                const bool ok = FULL || (col_ok[j] && row < g.M_real);
                dst[r] = ok ? ((const float *)g.out)[(size_t)row * g.ldo + col0 + j * 32] : 0.0f;
            }
        };
        load_unit(0, res[0]);
#pragma unroll
        for (int u = 0; u < WNT * WMT; ++u) {
            if (u + 1 < WNT * WMT) load_unit(u + 1, res[(u + 1) & 1]);
            const int i = u / WNT, j = u % WNT;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + i * 32 + (r & 3) + 8 * (r >> 2);
                if (!FULL && !(col_ok[j] && row < g.M_real)) continue;
                ((float *)g.out)[(size_t)row * g.ldo + col0 + j * 32] = (acc[i][j][r] + bv[j]) + res[u & 1][r];
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < WMT; ++i) {
        if constexpr (EPI == EPI_BIAS_RESID) {
        } else {
#pragma unroll
            for (int j = 0; j < WNT; ++j) {
                const int col = col0 + j * 32;
                if constexpr (EPI == EPI_BIAS_GELU) {
                    // two rows at a time: bias, round to the operand type (ggml's fp16 LUT input), packed tanh-GELU, round
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const int row = row0 + i * 32 + (r & 3) + 8 * (r >> 2);          // r even: rows `row` and `row + 1`
                        const typename Pair<T>::v2 xin = round_pair<T>(acc[i][j][r] + bv[j], acc[i][j][r + 1] + bv[j]);
                        const f32x2 y = gelu_tanh2(f32x2{(float)xin[0], (float)xin[1]});
                        const typename Pair<T>::v2 yo = round_pair<T>(y[0], y[1]);
                        if (FULL || (col_ok[j] && row < g.M_real)) ((T *)g.out)[(size_t)row * g.ldo + col] = yo[0];
                        if (FULL || (col_ok[j] && row + 1 < g.M_real)) ((T *)g.out)[(size_t)(row + 1) * g.ldo + col] = yo[1];
                    }
                    continue;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row0 + i * 32 + (r & 3) + 8 * (r >> 2);
                    if (!FULL && !(col_ok[j] && row < g.M_real)) continue;
                    const float v = acc[i][j][r] + bv[j];
                    if constexpr (EPI == EPI_BIAS) {
                        ((T *)g.out)[(size_t)row * g.ldo + col] = (T)v;
                    } else if constexpr (EPI == EPI_BIAS_GELU) {
                        ((T *)g.out)[(size_t)row * g.ldo + col] = (T)gelu_tanh(rnd<T>(v));
                    } else if constexpr (EPI == EPI_BIAS_F32) {
                        ((float *)g.out)[(size_t)row * g.ldo + col] = v;
                    } else {   // EPI_PATCH
                        const int b = row / g.tpi, t = row - b * g.tpi;
                        ((float *)g.out)[((size_t)row + b + 1) * g.ldo + col] = v + g.pos[(size_t)(t + 1) * g.ldo + col];
                    }
                }
            }
        }
    }
}

template <typename T, int EPI, int WMT, int WNT, int NWM, int NWN, int NS, int KS, bool DBG>
__global__ __launch_bounds__(NWM * NWN * 64, (NWM * NWN == 4 && WMT * WNT > 8) ? 1 : (NWM * NWN * 64) / 256) void gemm_ring_kernel(GemmArgs g) {
    constexpr int RBK = 16 * KS;                    // K per ring slot (KS k-steps of 16): 32 -> 64-B rows, 64 -> 128-B rows
    constexpr int ROWB = 2 * RBK;
    constexpr int NT = NWM * NWN * 64;              // threads: NWM x NWN waves
    constexpr int BM = NWM * WMT * 32, BN = NWN * WNT * 32;
    constexpr int A_BYTES = BM * ROWB, W_BYTES = BN * ROWB, SLOT_BYTES = A_BYTES + W_BYTES;
    constexpr int A_PIECES = BM * (ROWB / 16) / NT; // 16-B pieces per thread per slot
    constexpr int W_PIECES = BN * (ROWB / 16) / NT;
    constexpr int G = A_PIECES + W_PIECES;          // LDS-DMA instructions per thread per slot
    constexpr int PIECE_STRIDE = NT * 16;           // LDS bytes covered by one workgroup-wide DMA instruction
    typedef typename Elem<T>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / NWN, wn = wave % NWN;
    const int dbg = DBG ? g.dbg : 0;

    // ---- workgroup -> tile
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int ntm = g.M / BM, ntn = g.N_pad / BN;
    const int per_group = GROUP_M * ntn;
    const int grp = lid / per_group, within = lid - grp * per_group;
    const int gm = min(GROUP_M, ntm - grp * GROUP_M);
    const int tn = within / gm, tmi = grp * GROUP_M + (within - tn * gm);
    const int m0 = tmi * BM, n0 = tn * BN;

    // ---- per-thread DMA source offsets (elements) and wave-uniform LDS destinations
    const T *A = (const T *)g.A, *W = (const T *)g.W;
    int aoff[A_PIECES], woff[W_PIECES];
#pragma unroll
    for (int i = 0; i < A_PIECES; ++i) { int row, s; if (KS == 2) swz64_inv(i * NT + tid, row, s); else swz_inv(i * NT + tid, row, s); aoff[i] = (m0 + row) * g.lda + s * 8; }
#pragma unroll
    for (int i = 0; i < W_PIECES; ++i) { int row, s; if (KS == 2) swz64_inv(i * NT + tid, row, s); else swz_inv(i * NT + tid, row, s); woff[i] = (n0 + row) * g.ldw + s * 8; }
    auto issue = [&](int slot, int pos) {
        char *base = smem + pos * SLOT_BYTES + wave * 1024;
        const int k0 = slot * RBK;
#pragma unroll
        for (int i = 0; i < A_PIECES; ++i) __builtin_amdgcn_global_load_lds(GPTR(A + aoff[i] + k0), LPTR(base + i * PIECE_STRIDE), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < W_PIECES; ++i) __builtin_amdgcn_global_load_lds(GPTR(W + woff[i] + k0), LPTR(base + A_BYTES + i * PIECE_STRIDE), 16, 0, 0);
    };

    // ---- fragment read offsets within a slot: row = wave base + tile*32 + l31, 16-B slot = ks*2 + hh
    int a_rd[WMT][KS], w_rd[WNT][KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int i = 0; i < WMT; ++i) { const int r = wm * (WMT * 32) + i * 32 + l31; a_rd[i][ks] = KS == 2 ? swz64_byte(r, ks * 2 + hh) : swz_byte(r, ks * 2 + hh); }
#pragma unroll
        for (int j = 0; j < WNT; ++j) { const int r = wn * (WNT * 32) + j * 32 + l31; w_rd[j][ks] = A_BYTES + (KS == 2 ? swz64_byte(r, ks * 2 + hh) : swz_byte(r, ks * 2 + hh)); }
    }

    f32x16 acc[WMT][WNT];
#pragma unroll
    for (int i = 0; i < WMT; ++i)
#pragma unroll
        for (int j = 0; j < WNT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int nslots = g.K / RBK;
    // ---- prologue: NS-1 slots in flight, slots 0 and 1 landed
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) if (s < nslots) issue(s, s);
    {
        const int keep = max(0, min(NS - 1, nslots) - 2);      // slots allowed to stay in flight
        if (NS == 2) wait_vmcnt<0>();
        else if (NS > 3 && keep >= NS - 3) wait_vmcnt<(NS > 3 ? NS - 3 : 0) * G>(); else if (NS > 4 && keep == 1) wait_vmcnt<G>(); else wait_vmcnt<0>();
    }
    wg_barrier();

    v8 fa[2][WMT], fw[2][WNT];
    auto load_frags = [&](int buf, int pos, int ks) {
        const char *sb = smem + pos * SLOT_BYTES;
#pragma unroll
        for (int j = 0; j < WNT; ++j) fw[buf][j] = *(const v8 *)(sb + w_rd[j][ks]);
#pragma unroll
        for (int i = 0; i < WMT; ++i) fa[buf][i] = *(const v8 *)(sb + a_rd[i][ks]);
    };
    auto mma = [&](int buf) {
#pragma unroll
        for (int i = 0; i < WMT; ++i)
#pragma unroll
            for (int j = 0; j < WNT; ++j) acc[i][j] = Elem<T>::mfma(fa[buf][i], fw[buf][j], acc[i][j]);
    };
    if (NS > 2) load_frags(0, 0, 0);

    long long t_cyc = 0, t_real = 0;
    if (DBG && (dbg & 32)) { t_cyc = __builtin_readcyclecounter(); t_real = wall_clock64(); }
    if (DBG && (dbg & 2)) load_frags(1, 0, 1);
    int pos = 0;                                    // ring position of slot i
    for (int i = 0; i < nslots; ++i) {
        const int pos_next = (pos + 1 == NS) ? 0 : pos + 1;
        const int pos_fill = (pos == 0) ? NS - 1 : pos - 1;     // = (i + NS - 1) % NS, freed by the barrier that ended iteration i-1
        if (i + NS - 1 < nslots && !(dbg & 1)) issue(i + NS - 1, pos_fill);
        if (NS == 2 && !(dbg & 2)) load_frags(0, pos, 0);                      // double buffer: the slot only became visible at the barrier
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (!(dbg & 2)) {
                if (ks + 1 < KS) load_frags((ks + 1) & 1, pos, ks + 1);
#if RING_SCHED
                else if (NS > 2) load_frags((ks + 1) & 1, pos_next, 0);                     // unconditional: keeps the k-step one scheduling region (stale data past the end is never used)
#else
                else if (NS > 2 && i + 1 < nslots) load_frags((ks + 1) & 1, pos_next, 0);   // slot i+1 landed one iteration ago
#endif
            }
            if (!(dbg & 4)) mma(ks & 1);
#if RING_SCHED
            // MFMA first (its operands were read during the previous k-step), then one LDS read per MFMA
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
            for (int q = 0; q < (WMT + WNT) / RING_SCHED; ++q) { __builtin_amdgcn_sched_group_barrier(0x100, RING_SCHED, 0); __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); }
            if constexpr (WMT * WNT - 1 - (WMT + WNT) / RING_SCHED > 0) __builtin_amdgcn_sched_group_barrier(0x008, WMT * WNT - 1 - (WMT + WNT) / RING_SCHED, 0);
#endif
        }
        // slot i+2 must have landed before the next iteration's prefetch of it; later slots stay in flight
        if (!(dbg & 16)) {
            const int keep = min(i + NS - 1, nslots - 1) - (i + 2);            // slots allowed in flight (may be < 0)
            if (NS == 2) wait_vmcnt<0>();
            else if (NS > 3 && keep >= NS - 3) wait_vmcnt<(NS > 3 ? NS - 3 : 0) * G>();
            else if (NS > 4 && keep == 1) wait_vmcnt<G>();
            else wait_vmcnt<0>();
            wg_barrier();
        }
        pos = pos_next;
    }

    if (DBG && (dbg & 8)) {          // experiments: keep the accumulators and fragments alive, store nothing
        if (dbg & 4) { asm volatile("" ::"v"(fa[0][0]), "v"(fa[1][0]), "v"(fw[0][0]), "v"(fw[1][0])); }
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < WMT; ++i)
#pragma unroll
            for (int j = 0; j < WNT; ++j) s += acc[i][j][0] + acc[i][j][15];
        if (s == 1234.5678f) ((float *)g.out)[0] = s;
        if ((dbg & 32) && tid == 0) {   // per-block timeline: K-loop start/end on the 100 MHz wall clock, shader cycles
            long long *d = (long long *)g.pos + (size_t)bid * 4;
            d[0] = t_real; d[1] = wall_clock64(); d[2] = __builtin_readcyclecounter() - t_cyc; d[3] = xcd;
        }
        return;
    }
    const bool full = (m0 + BM <= g.M_real) && (n0 + BN <= g.N);
    float bv[WNT];
    load_bias<WNT>(g, n0 + wn * (WNT * 32) + l31, bv);
    if (full) epilogue<T, EPI, WMT, WNT, true>(g, acc, m0 + wm * (WMT * 32) + 4 * hh, n0 + wn * (WNT * 32) + l31, bv);
    else epilogue<T, EPI, WMT, WNT, false>(g, acc, m0 + wm * (WMT * 32) + 4 * hh, n0 + wn * (WNT * 32) + l31, bv);
}

// ---- persistent variant: one workgroup per CU walks tiles v = bid, bid + grid, ... and keeps ONE ring running
// across tile boundaries: the slots of the next tile are already in flight (and its first fragments in registers)
// while the epilogue of the current tile stores, so neither the launch gap nor the cold prologue of a fresh
// workgroup is paid per tile (measured 6.6 us of 27 us per 256x256x768 tile in the one-tile-per-workgroup kernel).
// vmcnt counts the epilogue's stores as well as the DMA loads; loads retire in order among themselves, so a
// counted wait stays correct (only stricter) while stores are outstanding.  To keep the stores off the critical
// path every slot in flight is drained BEFORE the epilogue (they are 1-4 slots old) and the first two slots of
// the next tile then need no wait at all.
template <typename T, int EPI, int WMT, int WNT, int NWM, int NWN, int NS, int KS>
__global__ __launch_bounds__(NWM * NWN * 64, (NWM * NWN * 64) / 256) void gemm_stream_kernel(GemmArgs g) {
    constexpr int RBK = 16 * KS, ROWB = 2 * RBK, NT = NWM * NWN * 64;
    constexpr int BM = NWM * WMT * 32, BN = NWN * WNT * 32;
    constexpr int A_BYTES = BM * ROWB, W_BYTES = BN * ROWB, SLOT_BYTES = A_BYTES + W_BYTES;
    constexpr int A_PIECES = BM * (ROWB / 16) / NT, W_PIECES = BN * (ROWB / 16) / NT;
    constexpr int G = A_PIECES + W_PIECES, PIECE_STRIDE = NT * 16;
    static_assert(NS >= 4 && KS == 2, "stream kernel: 64-byte rows, ring of >= 4 slots");
    typedef typename Elem<T>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / NWN, wn = wave % NWN;

    // ---- tile walk: virtual workgroup id v keeps v % 8 == bid % 8 (same XCD), then the XCD-contiguous GROUP_M raster
    const int ntm = g.M / BM, ntn = g.N_pad / BN, ntiles = ntm * ntn;
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int my_tiles = (ntiles - bid + nblk - 1) / nblk;
    const int q8 = ntiles >> 3, r8 = ntiles & 7, xcd = bid & 7;
    const int lid_base = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8);
    auto tile_origin = [&](int round, int &m0, int &n0) {
        const int v = bid + round * nblk;
        const int lid = lid_base + (v >> 3);
        const int per_group = GROUP_M * ntn;
        const int grp = lid / per_group, within = lid - grp * per_group;
        const int gm = min(GROUP_M, ntm - grp * GROUP_M);
        const int tn = within / gm;
        m0 = (grp * GROUP_M + (within - tn * gm)) * BM; n0 = tn * BN;
    };

    const T *A = (const T *)g.A, *W = (const T *)g.W;
    int aoff[A_PIECES], woff[W_PIECES];
#pragma unroll
    for (int i = 0; i < A_PIECES; ++i) { int row, sl; swz64_inv(i * NT + tid, row, sl); aoff[i] = row * g.lda + sl * 8; }
#pragma unroll
    for (int i = 0; i < W_PIECES; ++i) { int row, sl; swz64_inv(i * NT + tid, row, sl); woff[i] = row * g.ldw + sl * 8; }

    const int nslots = g.K / RBK;
    const int S = my_tiles * nslots;                // slots this workgroup streams through its ring
    // issue side runs NS-1 slots ahead of the consume side, across tile boundaries
    int is_round = 0, is_slot = 0, is_pos = 0, issued = 0, is_a = 0, is_w = 0;
    if (my_tiles > 0) { int m0, n0; tile_origin(0, m0, n0); is_a = m0 * g.lda; is_w = n0 * g.ldw; }
    auto issue_next = [&]() {
        char *base = smem + is_pos * SLOT_BYTES + wave * 1024;
        const T *Ab = A + is_a + is_slot * RBK, *Wb = W + is_w + is_slot * RBK;
#pragma unroll
        for (int i = 0; i < A_PIECES; ++i) __builtin_amdgcn_global_load_lds(GPTR(Ab + aoff[i]), LPTR(base + i * PIECE_STRIDE), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < W_PIECES; ++i) __builtin_amdgcn_global_load_lds(GPTR(Wb + woff[i]), LPTR(base + A_BYTES + i * PIECE_STRIDE), 16, 0, 0);
        ++issued;
        is_pos = (is_pos + 1 == NS) ? 0 : is_pos + 1;
        if (++is_slot == nslots) {
            is_slot = 0; ++is_round;
            if (is_round < my_tiles) { int m0, n0; tile_origin(is_round, m0, n0); is_a = m0 * g.lda; is_w = n0 * g.ldw; }
        }
    };

    int a_rd[WMT][KS], w_rd[WNT][KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int i = 0; i < WMT; ++i) a_rd[i][ks] = swz64_byte(wm * (WMT * 32) + i * 32 + l31, ks * 2 + hh);
#pragma unroll
        for (int j = 0; j < WNT; ++j) w_rd[j][ks] = A_BYTES + swz64_byte(wn * (WNT * 32) + j * 32 + l31, ks * 2 + hh);
    }
    v8 fa[2][WMT], fw[2][WNT];
    auto load_frags = [&](int buf, int pos, int ks) {
        const char *sb = smem + pos * SLOT_BYTES;
#pragma unroll
        for (int j = 0; j < WNT; ++j) fw[buf][j] = *(const v8 *)(sb + w_rd[j][ks]);
#pragma unroll
        for (int i = 0; i < WMT; ++i) fa[buf][i] = *(const v8 *)(sb + a_rd[i][ks]);
    };
    f32x16 acc[WMT][WNT];
    auto mma = [&](int buf) {
#pragma unroll
        for (int i = 0; i < WMT; ++i)
#pragma unroll
            for (int j = 0; j < WNT; ++j) acc[i][j] = Elem<T>::mfma(fa[buf][i], fw[buf][j], acc[i][j]);
    };
    if (S == 0) return;

    // ---- prologue (once per workgroup): NS-1 slots in flight, slots 0 and 1 landed, first fragments in registers
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) if (issued < S) issue_next();
    if (S >= NS - 1) wait_vmcnt<(NS - 3) * G>(); else wait_vmcnt<0>();
    wg_barrier();
    load_frags(0, 0, 0);

    int pos = 0, gi = 0;                            // ring position / global index of the slot being consumed
    for (int round = 0; round < my_tiles; ++round) {
        // the tile's bias is fetched now and is known complete at the drain wait below (a compiler-visible s_waitcnt:
        // no load may stay pending into the next round, or hipcc guards the loop's register reuse with vmcnt(0))
        int m0, n0; tile_origin(round, m0, n0);
        float bv[WNT];
        load_bias<WNT>(g, n0 + wn * (WNT * 32) + l31, bv);
#pragma unroll
        for (int i = 0; i < WMT; ++i)
#pragma unroll
            for (int j = 0; j < WNT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
        for (int i = 0; i < nslots; ++i, ++gi) {
            const int pos_next = (pos + 1 == NS) ? 0 : pos + 1;
            if (issued < S) issue_next();           // slot gi+NS-1 into the position freed by the barrier that ended slot gi-1
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks + 1 < KS) load_frags((ks + 1) & 1, pos, ks + 1);
                else load_frags((ks + 1) & 1, pos_next, 0);      // slot gi+1 (possibly the next tile's first) landed one iteration ago
                mma(ks & 1);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
                for (int q = 0; q < WMT + WNT; ++q) { __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); }
                __builtin_amdgcn_sched_group_barrier(0x008, WMT * WNT - 1 - (WMT + WNT), 0);
            }
            // slot gi+2 must have landed before the next iteration prefetches its fragments
            const bool tile_end = (i == nslots - 1);
            if (tile_end) __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0) drain: nothing in flight during the epilogue's stores
            else if (round > 0 && i < NS - 3) { /* slots gi+2 <= first NS-1 of this tile: drained before the previous epilogue */ }
            else {
                const int keep = min(gi + NS - 1, S - 1) - (gi + 2);         // slots allowed to stay in flight
                if (keep >= NS - 3) wait_vmcnt<(NS - 3) * G>();
                else if (NS > 4 && keep == 1) wait_vmcnt<G>();
                else wait_vmcnt<0>();
            }
            wg_barrier();
            pos = pos_next;
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0) again, unconditionally and visible to hipcc (free: the last slot's wait drained everything)
        const bool full = (m0 + BM <= g.M_real) && (n0 + BN <= g.N);
        if (full) epilogue<T, EPI, WMT, WNT, true>(g, acc, m0 + wm * (WMT * 32) + 4 * hh, n0 + wn * (WNT * 32) + l31, bv);
        else epilogue<T, EPI, WMT, WNT, false>(g, acc, m0 + wm * (WMT * 32) + 4 * hh, n0 + wn * (WNT * 32) + l31, bv);
    }
}

// ---- configurations: cfg = WMT*100 + NWN*10 + NS
struct RingCfg { int wmt, nwn, ns, ks, wnt, nwm; };
static bool parse_cfg(int cfg, RingCfg &c) {
    c.ks = 2; c.wnt = 2; c.nwm = 2; c.ns = 5;
    switch (cfg) {
    case 445: case 945: c.wmt = 4; c.nwn = 4; return true;      // 256x256, one workgroup per tile / persistent stream kernel
    case 245: c.wmt = 2; c.nwn = 4; return true;                // 128x256
    case 122: c.wmt = 1; c.nwn = 2; return true;                // skinny: 4 waves, tile 64x128, 60 KiB ring, 2 workgroups/CU (latency of tiny batches)
    default: return false;
    }
}

template <typename T, int EPI, int WMT, int WNT, int NWM, int NWN, int NS, int KS, bool DBG>
static hipError_t launch_ring_inst(const GemmArgs &a, hipStream_t stream, bool prepare) {
    constexpr int BM = NWM * WMT * 32, BN = NWN * WNT * 32;
    constexpr int lds = NS * (BM + BN) * 32 * KS;
    if (prepare || DBG) {       // once per device (tuning_for_device); the ablation builds set it on every launch
        hipError_t e = hipFuncSetAttribute((const void *)gemm_ring_kernel<T, EPI, WMT, WNT, NWM, NWN, NS, KS, DBG>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (prepare) return e;
    }
    const int grid = (a.M / BM) * (a.N_pad / BN);
    hipLaunchKernelGGL((gemm_ring_kernel<T, EPI, WMT, WNT, NWM, NWN, NS, KS, DBG>), dim3(grid), dim3(NWM * NWN * 64), lds, stream, a);
    return hipGetLastError();
}

template <typename T, int EPI>
static hipError_t launch_stream_inst(const GemmArgs &a, int n_cu, hipStream_t stream, bool prepare) {
    constexpr int WMT = 4, WNT = 2, NWM = 2, NWN = 4, NS = 5, KS = 2;
    constexpr int BM = NWM * WMT * 32, BN = NWN * WNT * 32, lds = NS * (BM + BN) * 32 * KS;
    if (prepare) return hipFuncSetAttribute((const void *)gemm_stream_kernel<T, EPI, WMT, WNT, NWM, NWN, NS, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    n_cu &= ~7;                                      // the tile walk keeps a workgroup on one XCD: grid must be a multiple of 8
    if (n_cu <= 0) n_cu = 256;
    const int ntiles = (a.M / BM) * (a.N_pad / BN);
    const int grid = ntiles < n_cu ? ntiles : n_cu;
    hipLaunchKernelGGL((gemm_stream_kernel<T, EPI, WMT, WNT, NWM, NWN, NS, KS>), dim3(grid), dim3(NWM * NWN * 64), lds, stream, a);
    return hipGetLastError();
}

template <typename T, int EPI, bool DBG>
static hipError_t launch_ring_e(const GemmArgs &a, int cfg, int n_cu, hipStream_t stream, bool prepare) {
    switch (cfg) {
    case 945: if constexpr (!DBG) return launch_stream_inst<T, EPI>(a, n_cu, stream, prepare); else return hipErrorInvalidValue;
    case 445: return launch_ring_inst<T, EPI, 4, 2, 2, 4, 5, 2, DBG>(a, stream, prepare);
    case 245: return launch_ring_inst<T, EPI, 2, 2, 2, 4, 5, 2, DBG>(a, stream, prepare);
    case 122: return launch_ring_inst<T, EPI, 1, 2, 2, 2, 5, 2, DBG>(a, stream, prepare);
    default: return hipErrorInvalidValue;
    }
}
template <typename T>
static hipError_t launch_ring_t(const GemmArgs &a, int epi, int cfg, int n_cu, hipStream_t stream, bool prepare) {
    if (a.dbg) return epi == EPI_BIAS ? launch_ring_e<T, EPI_BIAS, true>(a, cfg, n_cu, stream, prepare) : hipErrorInvalidValue;
    switch (epi) {
    case EPI_BIAS: return launch_ring_e<T, EPI_BIAS, false>(a, cfg, n_cu, stream, prepare);
    case EPI_BIAS_GELU: return launch_ring_e<T, EPI_BIAS_GELU, false>(a, cfg, n_cu, stream, prepare);
    case EPI_BIAS_RESID: return launch_ring_e<T, EPI_BIAS_RESID, false>(a, cfg, n_cu, stream, prepare);
    case EPI_BIAS_F32: return launch_ring_e<T, EPI_BIAS_F32, false>(a, cfg, n_cu, stream, prepare);
    case EPI_PATCH: return launch_ring_e<T, EPI_PATCH, false>(a, cfg, n_cu, stream, prepare);
    default: return hipErrorInvalidValue;
    }
}

bool gemm_ring_supports(const GemmArgs &a, int cfg) {
    RingCfg c;
    if (!parse_cfg(cfg, c)) return false;
    if (cfg == 945 && a.K < 16 * c.ks * c.ns) return false;
    return a.M % (c.nwm * c.wmt * 32) == 0 && a.N_pad % (c.nwn * c.wnt * 32) == 0 && a.K % (16 * c.ks) == 0 && a.K >= 32 * c.ks;
}

hipError_t launch_gemm_ring(const Tuning &t, int dtype, int epi, const GemmArgs &a0, int cfg, hipStream_t stream, bool prepare) {
    if (prepare) return dtype == DT_F16 ? launch_ring_t<_Float16>(a0, epi, cfg, t.n_cu, stream, true) : launch_ring_t<__bf16>(a0, epi, cfg, t.n_cu, stream, true);
    if (!gemm_ring_supports(a0, cfg)) return hipErrorInvalidValue;
    const int dbg = t.gemm_dbg;
    GemmArgs a = a0; a.dbg = dbg;
    if (dbg & 32) {      // experiment mode: collect and print a per-block timeline (synchronous)
        RingCfg c; parse_cfg(cfg, c);
        const int nwg = (a.M / (c.nwm * c.wmt * 32)) * (a.N_pad / (c.nwn * c.wnt * 32));
        long long *buf = nullptr;
        if (hipHostMalloc((void **)&buf, (size_t)nwg * 32, 0) != hipSuccess) return hipErrorOutOfMemory;
        a.pos = (const float *)buf;
        hipError_t e = dtype == DT_F16 ? launch_ring_t<_Float16>(a, epi, cfg, t.n_cu, stream, false) : launch_ring_t<__bf16>(a, epi, cfg, t.n_cu, stream, false);
        (void)hipDeviceSynchronize();
        long long t0 = buf[0], t1 = 0; double sum = 0, sumc = 0;
        for (int b = 0; b < nwg; ++b) { t0 = std::min(t0, buf[b * 4]); t1 = std::max(t1, buf[b * 4 + 1]); sum += buf[b * 4 + 1] - buf[b * 4]; sumc += buf[b * 4 + 2]; }
        fprintf(stderr, "[ring dbg] cfg %d: %d blocks, span %.1f us, mean K-loop %.1f us (%.0f shader cycles, %.0f MHz)\n", cfg, nwg, (t1 - t0) / 100.0, sum / nwg / 100.0, sumc / nwg, sumc / sum * 100.0);
        (void)hipHostFree(buf);
        return e;
    }
    return dtype == DT_F16 ? launch_ring_t<_Float16>(a, epi, cfg, t.n_cu, stream, false) : launch_ring_t<__bf16>(a, epi, cfg, t.n_cu, stream, false);
}

}  // namespace vitx

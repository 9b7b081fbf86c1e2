// gemm_ring.hip -- the main MFMA GEMM of the ViT forward path for gfx950.
//
//   C[M][N] = A[M][K] . W[N][K]^T   (+ fused epilogue), A/W fp16 or bf16, f32 accumulate.
//
// Design (MI355X-first, see DESIGN.md "GEMM"):
//   * waves are laid out 2(M) x 4(N); each wave owns (WMT*32) x 64 of C as 2 WMT x 4 accumulators of
//     v_mfma_f32_16x16x32 (11 % less energy per flop than 32x32x16 on this power-capped part, see gemm_pp.hip):
//     8 waves, tile 256x256 (WMT=4, cfg 445/945) or 128x256 (WMT=2, cfg 245: short M and
//     the remainder rows of a launch), one workgroup per CU, ring of 5 slots (160 / 120 KiB LDS).
//   * cfg 945 (default for the wide tile) is PERSISTENT: one workgroup per CU walks its tiles and keeps
//     the ring running across tile boundaries (gemm_stream_kernel below).
//   * K is consumed in slots of 32: an LDS ring of NS slots, each [A: BM x 32 | W: BN x 32] halves.
//     Slots are filled by LDS-DMA (global_load_lds dwordx4) issued NS-1 slots ahead and retired with
//     COUNTED s_waitcnt vmcnt(N) + one raw s_barrier per slot, so loads stay in flight across
//     barriers (the compiler's __syncthreads() would drain them).
//   * a slot is ONE 32-deep k-step, consumed in two half-steps (upper / lower half of the wave's rows x all of its
//     columns).  Fragment registers are double buffered: the A fragments of the other half (and, under the second
//     half-step, the W and first A fragments of the next, already-landed slot) are read under the MFMAs of the current
//     half-step -- MFMA FIRST, then one read per MFMA (sched_group_barrier), so the lgkmcnt(0) in front of a half-step
//     waits on reads that are 2+ MFMAs old instead of on the reads just issued (+5-7 %).  The W buffer alternates with
//     the slot parity: the slot loop is unrolled by two (K % 64 == 0).
//   * 64-byte LDS rows, 4 rows per 256-B bank line, 16-B slots XOR-ed with (line & 15): every
//     ds_read_b128 lane group hits 16 distinct slots (conflict-free); the LDS image itself is
//     lane-linear (DMA requirement), the permutation is applied to the per-lane global source address.
//   * workgroup -> tile map: XCD-contiguous ids, then GROUP_M x n blocks so the tiles co-resident on
//     one XCD share A and W panels in that XCD's 4 MiB L2.
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "device_common.h"
#include "kernels.h"
#include "epilogue16.h"

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

// "1 MFMA, then one LDS read per MFMA, then the rest": R reads under MF MFMAs
template <int R, int MF>
__device__ __forceinline__ void sched_half_step() {
    constexpr int n = R < MF - 1 ? R : MF - 1;
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
    for (int q = 0; q < n; ++q) { __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); }
    if constexpr (R > n) __builtin_amdgcn_sched_group_barrier(0x100, R - n, 0);
    if constexpr (MF - 1 - n > 0) __builtin_amdgcn_sched_group_barrier(0x008, MF - 1 - n, 0);
}
typedef std::integral_constant<int, 0> RI0; typedef std::integral_constant<int, 1> RI1;

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

    static_assert(KS == 2 && NS > 2, "ring kernel: 64-byte rows (one 32-deep k-step per slot), ring of >= 3 slots");
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g4 = lane >> 4;
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

    // ---- fragment read offsets within a slot: row = wave base + tile*16 + l15, 16-B slot = g4 (which 8 of the slot's 32 k values)
    int a_rd[2 * WMT], w_rd[2 * WNT];
#pragma unroll
    for (int t = 0; t < 2 * WMT; ++t) a_rd[t] = swz64_byte(wm * (WMT * 32) + t * 16 + l15, g4);
#pragma unroll
    for (int u = 0; u < 2 * WNT; ++u) w_rd[u] = A_BYTES + swz64_byte(wn * (WNT * 32) + u * 16 + l15, g4);

    f32x4 acc[2 * WMT][2 * WNT];
#pragma unroll
    for (int t = 0; t < 2 * WMT; ++t)
#pragma unroll
        for (int u = 0; u < 2 * WNT; ++u) acc[t][u] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

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

    // fa[h]: A fragments of row half h (tiles h * WMT ..); fw[p]: all W fragments of a slot of parity p
    v8 fa[2][WMT], fw[2][2 * WNT];
    auto load_a = [&](int h, int pos) {
        const char *sb = smem + pos * SLOT_BYTES;
#pragma unroll
        for (int t = 0; t < WMT; ++t) fa[h][t] = *(const v8 *)(sb + a_rd[h * WMT + t]);
    };
    auto load_w = [&](int p, int pos) {
        const char *sb = smem + pos * SLOT_BYTES;
#pragma unroll
        for (int u = 0; u < 2 * WNT; ++u) fw[p][u] = *(const v8 *)(sb + w_rd[u]);
    };
    auto mma = [&](int h, int p) {
#pragma unroll
        for (int t = 0; t < WMT; ++t)
#pragma unroll
            for (int u = 0; u < 2 * WNT; ++u) acc[h * WMT + t][u] = Elem<T>::mfma16(fw[p][u], fa[h][t], acc[h * WMT + t][u]);
    };
    load_w(0, 0); load_a(0, 0);

    long long t_cyc = 0, t_real = 0;
    if (DBG && (dbg & 32)) { t_cyc = __builtin_readcyclecounter(); t_real = wall_clock64(); }
    if (DBG && (dbg & 2)) { load_a(1, 0); load_w(1, 0); }
    int pos = 0;                                    // ring position of slot i
    auto slot = [&](auto pc, int i) {
        constexpr int P = decltype(pc)::value;      // slot parity = which W fragment buffer it uses
        const int pos_next = (pos + 1 == NS) ? 0 : pos + 1;
        const int pos_fill = (pos == 0) ? NS - 1 : pos - 1;     // = (i + NS - 1) % NS, freed by the barrier that ended iteration i-1
        if (i + NS - 1 < nslots && !(dbg & 1)) issue(i + NS - 1, pos_fill);
        // upper half of the rows; the lower half's A fragments are read underneath
        if (!(dbg & 2)) load_a(1, pos);
        if (!(dbg & 4)) mma(0, P);
        sched_half_step<WMT, WMT * 2 * WNT>();
        // lower half; slot i+1 landed one iteration ago (stale data past the end is never used): its W and first A fragments
        if (!(dbg & 2)) { load_w(P ^ 1, pos_next); load_a(0, pos_next); }
        if (!(dbg & 4)) mma(1, P);
        sched_half_step<WMT + 2 * WNT, WMT * 2 * WNT>();
        // slot i+2 must have landed before the next iteration's prefetch of it; later slots stay in flight
        if (!(dbg & 16)) {
            const int keep = min(i + NS - 1, nslots - 1) - (i + 2);            // slots allowed in flight (may be < 0)
            if (NS > 3 && keep >= NS - 3) wait_vmcnt<(NS > 3 ? NS - 3 : 0) * G>();
            else if (NS > 4 && keep == 1) wait_vmcnt<G>();
            else wait_vmcnt<0>();
            wg_barrier();
        }
        pos = pos_next;
    };
    for (int i = 0; i < nslots; i += 2) { slot(RI0{}, i); slot(RI1{}, i + 1); }     // K % 64 == 0 (gemm_ring_supports)

    if (DBG && (dbg & 8)) {          // experiments: keep the accumulators and fragments alive, store nothing
        if (dbg & 4) { asm volatile("" ::"v"(fa[0][0]), "v"(fa[1][0]), "v"(fw[0][0]), "v"(fw[1][0])); }
        float s = 0.0f;
#pragma unroll
        for (int t = 0; t < 2 * WMT; ++t)
#pragma unroll
            for (int u = 0; u < 2 * WNT; ++u) s += acc[t][u][0] + acc[t][u][3];
        if (s == 1234.5678f) ((float *)g.out)[0] = s;
        if ((dbg & 32) && tid == 0) {   // per-block timeline: K-loop start/end on the 100 MHz wall clock, shader cycles
            long long *d = (long long *)g.pos + (size_t)bid * 4;
            d[0] = t_real; d[1] = wall_clock64(); d[2] = __builtin_readcyclecounter() - t_cyc; d[3] = xcd;
        }
        return;
    }
    // the last slot's barrier has passed and nothing is in flight: the ring is free, every wave takes 4 KiB of it as its epilogue patch
    static_assert(WNT == 2, "epilogue16_tile: 64 columns per wave");
    const bool full = (m0 + BM <= g.M_real) && (n0 + BN <= g.N);
    epilogue16_tile<T, EPI, WMT>(g, acc, full, m0, n0, wm * (WMT * 32), wn * (WNT * 32), smem + wave * 4096, lane);
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

    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g4 = lane >> 4;
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

    int a_rd[2 * WMT], w_rd[2 * WNT];
#pragma unroll
    for (int t = 0; t < 2 * WMT; ++t) a_rd[t] = swz64_byte(wm * (WMT * 32) + t * 16 + l15, g4);
#pragma unroll
    for (int u = 0; u < 2 * WNT; ++u) w_rd[u] = A_BYTES + swz64_byte(wn * (WNT * 32) + u * 16 + l15, g4);
    v8 fa[2][WMT], fw[2][2 * WNT];                   // as in gemm_ring_kernel: A per row half, W per slot parity
    auto load_a = [&](int h, int pos) {
        const char *sb = smem + pos * SLOT_BYTES;
#pragma unroll
        for (int t = 0; t < WMT; ++t) fa[h][t] = *(const v8 *)(sb + a_rd[h * WMT + t]);
    };
    auto load_w = [&](int p, int pos) {
        const char *sb = smem + pos * SLOT_BYTES;
#pragma unroll
        for (int u = 0; u < 2 * WNT; ++u) fw[p][u] = *(const v8 *)(sb + w_rd[u]);
    };
    f32x4 acc[2 * WMT][2 * WNT];
    auto mma = [&](int h, int p) {
#pragma unroll
        for (int t = 0; t < WMT; ++t)
#pragma unroll
            for (int u = 0; u < 2 * WNT; ++u) acc[h * WMT + t][u] = Elem<T>::mfma16(fw[p][u], fa[h][t], acc[h * WMT + t][u]);
    };
    if (S == 0) return;

    // ---- prologue (once per workgroup): NS-1 slots in flight, slots 0 and 1 landed, first fragments in registers
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) if (issued < S) issue_next();
    if (S >= NS - 1) wait_vmcnt<(NS - 3) * G>(); else wait_vmcnt<0>();
    wg_barrier();
    load_w(0, 0); load_a(0, 0);

    int pos = 0, gi = 0;                            // ring position / global index of the slot being consumed
    int round = 0;
    auto slot = [&](auto pc, int i) {
        constexpr int P = decltype(pc)::value;
        const int pos_next = (pos + 1 == NS) ? 0 : pos + 1;
        if (issued < S) issue_next();               // slot gi+NS-1 into the position freed by the barrier that ended slot gi-1
        load_a(1, pos);
        mma(0, P);
        sched_half_step<WMT, WMT * 2 * WNT>();
        load_w(P ^ 1, pos_next); load_a(0, pos_next);            // slot gi+1 (possibly the next tile's first) landed one iteration ago
        mma(1, P);
        sched_half_step<WMT + 2 * WNT, WMT * 2 * WNT>();
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
        pos = pos_next; ++gi;
    };
    for (; round < my_tiles; ++round) {
        int m0, n0; tile_origin(round, m0, n0);
#pragma unroll
        for (int t = 0; t < 2 * WMT; ++t)
#pragma unroll
            for (int u = 0; u < 2 * WNT; ++u) acc[t][u] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        for (int i = 0; i < nslots; i += 2) { slot(RI0{}, i); slot(RI1{}, i + 1); }     // nslots is even: the parity restarts with every tile
        __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0) again, unconditionally and visible to hipcc (free: the last slot's wait drained everything)
        // the ring already holds the next tile's first slots: the epilogue patches (4 KiB per wave) live behind it
        const bool full = (m0 + BM <= g.M_real) && (n0 + BN <= g.N);
        epilogue16_tile<T, EPI, WMT>(g, acc, full, m0, n0, wm * (WMT * 32), wn * (WNT * 32), smem + NS * SLOT_BYTES + wave * 4096, lane);
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
    constexpr int WMT = 4, WNT = 2, NWM = 2, NWN = 4, NS = 4, KS = 2;      // 4 slots of 32 KiB + 8 epilogue patches of 4 KiB = 160 KiB
    constexpr int BM = NWM * WMT * 32, BN = NWN * WNT * 32, lds = NS * (BM + BN) * 32 * KS + NWM * NWN * 4096;
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
#ifdef VITX_LAB      // ablation build of the ring kernel (tools/): garbage results by design, never in the product library
    if (a.dbg) return epi == EPI_BIAS ? launch_ring_e<T, EPI_BIAS, true>(a, cfg, n_cu, stream, prepare) : hipErrorInvalidValue;
#endif
    switch (epi) {
    case EPI_BIAS: return launch_ring_e<T, EPI_BIAS, false>(a, cfg, n_cu, stream, prepare);
    case EPI_BIAS_GELU: return launch_ring_e<T, EPI_BIAS_GELU, false>(a, cfg, n_cu, stream, prepare);
    case EPI_BIAS_RESID: return launch_ring_e<T, EPI_BIAS_RESID, false>(a, cfg, n_cu, stream, prepare);
    case EPI_BIAS_F32: return launch_ring_e<T, EPI_BIAS_F32, false>(a, cfg, n_cu, stream, prepare);
    case EPI_PATCH: return launch_ring_e<T, EPI_PATCH, false>(a, cfg, n_cu, stream, prepare);
    case EPI_BIAS_HILO: return launch_ring_e<T, EPI_BIAS_HILO, false>(a, cfg, n_cu, stream, prepare);
    default: return hipErrorInvalidValue;
    }
}

bool gemm_ring_supports(const GemmArgs &a, int cfg) {
    RingCfg c;
    if (!parse_cfg(cfg, c)) return false;
    if (cfg == 945 && a.K < 16 * c.ks * 4) return false;      // the stream kernel's ring has 4 slots
    return a.M % (c.nwm * c.wmt * 32) == 0 && a.N_pad % (c.nwn * c.wnt * 32) == 0 && a.K % (32 * c.ks) == 0 && a.K >= 32 * c.ks;     // slot pairs: K % 64 == 0
}

hipError_t launch_gemm_ring(const Tuning &t, int dtype, int epi, const GemmArgs &a0, int cfg, hipStream_t stream, bool prepare) {
    if (prepare) return dtype == DT_F16 ? launch_ring_t<_Float16>(a0, epi, cfg, t.n_cu, stream, true) : launch_ring_t<__bf16>(a0, epi, cfg, t.n_cu, stream, true);
    if (!gemm_ring_supports(a0, cfg)) return hipErrorInvalidValue;
    GemmArgs a = a0; a.dbg = 0;
#ifdef VITX_LAB
    const int dbg = t.gemm_dbg;
    a.dbg = dbg;
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
#endif
    return dtype == DT_F16 ? launch_ring_t<_Float16>(a, epi, cfg, t.n_cu, stream, false) : launch_ring_t<__bf16>(a, epi, cfg, t.n_cu, stream, false);
}

}  // namespace vitx

// epilogue16.h -- fused epilogue shared by every GEMM kernel, for accumulators of v_mfma_f32_16x16x32_{f16,bf16}.
//
// All GEMM families compute C = A . W^T with SWAPPED products, acc = mfma16(W fragment, A fragment, acc): the 16 x 16 result tile
// then has, in lane (l15 = lane & 15, g4 = lane >> 4), register e = C[row l15][column 4 g4 + e] -- four CONSECUTIVE columns of one
// row, so a lane stores 8 bytes (f16 / bf16 outputs) or 16 bytes (f32 outputs) at a time and the residual read-modify-write is one
// dwordx4 load + one dwordx4 store.  Every family consumes K in the same order with the same instruction, so a given output
// element is the same bits whichever kernel the batch size selects (tests/test_gpu_parity_r02.py).
//
// Epilogues (reference: /root/reference/vit.cpp): EPI_BIAS qkv (:820-823); EPI_BIAS_GELU fc1 + ggml_gelu through the fp16
// table (:888-893; gelu_out_pair: F16 rounds the value before and after the activation as the table does, BF16 only after); EPI_BIAS_RESID proj / fc2 +
// residual add in f32 (:868-873, :899-902); EPI_BIAS_F32 classifier head (:917-922); EPI_PATCH patch embedding + position
// embedding, patch row -> token row (:774-800).
#pragma once
#include "device_common.h"
#include "kernels.h"

namespace vitx {

// EPI_BIAS_HILO: lo = round((v - hi) * 2048) for two adjacent outputs, hi = the already rounded pair
template <typename T> __device__ __forceinline__ typename Pair<T>::v2 hilo_lo_pair(float v0, float v1, typename Pair<T>::v2 hi) {
    return round_pair<T>((v0 - (float)hi[0]) * kHiLoScale, (v1 - (float)hi[1]) * kHiLoScale);
}

// How an epilogue takes an accumulator tile: plain values (every kernel but one), or -- gemm_w4.hip -- tiles pinned to physical accumulator
// registers behind inline assembly, copied to VGPRs at the point of use (AccPhys there).  idx = 8 * (16-row block) + column tile + idx0.
struct AccPlain { static __device__ __forceinline__ f32x4 get(const f32x4 &a, int) { return a; } };

// acc[t][u]: tile rows row0 + 16 t (row0 already includes l15), columns col0 + 16 u .. + 3 (col0 already includes 4 g4).
// FULL: the whole workgroup tile is inside [0, M_real) x [0, N) -- no bounds checks, vector bias loads.  Otherwise every element is
// checked on its own (N need not be a multiple of 4: the classifier head has as many columns as the model has classes).
template <typename T, int EPI, int TM, int TN, bool FULL, typename ACC = AccPlain>
__device__ __forceinline__ void epilogue16(const GemmArgs &g, f32x4 (&acc)[TM][TN], int row0, int col0, int idx0 = 0) {
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
        if constexpr (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_HILO) {
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                const int row = row0 + t * 16;
                if (!FULL && row >= g.M_real) continue;
                const f32x4 v = ACC::get(acc[t][u], t * 8 + u + idx0) + bv;
                v2 p0, p1;
                if constexpr (EPI == EPI_BIAS_GELU) { p0 = gelu_out_pair<T>(v[0], v[1]); p1 = gelu_out_pair<T>(v[2], v[3]); }
                else { p0 = round_pair<T>(v[0], v[1]); p1 = round_pair<T>(v[2], v[3]); }
                T *o = (T *)g.out + (size_t)row * g.ldo + c;
                if (vec) *(v4 *)o = v4{p0[0], p0[1], p1[0], p1[1]};
                else {
                    const T s[4] = {p0[0], p0[1], p1[0], p1[1]};
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (c + e < g.N) o[e] = s[e];
                }
                if constexpr (EPI == EPI_BIAS_HILO) {        // the lo plane: what the 16-bit rounding above dropped, scaled into the type's normal range
                    const v2 l0 = hilo_lo_pair<T>(v[0], v[1], p0), l1 = hilo_lo_pair<T>(v[2], v[3], p1);
                    T *ol = o + g.hilo_off;
                    if (vec) *(v4 *)ol = v4{l0[0], l0[1], l1[0], l1[1]};
                    else {
                        const T s[4] = {l0[0], l0[1], l1[0], l1[1]};
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (c + e < g.N) ol[e] = s[e];
                    }
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
                f32x4 r = ACC::get(acc[t][u], t * 8 + u + idx0) + bv;
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

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// The patch is written and read back by the same wave through different vector types: a compiler-level barrier keeps those
// accesses in program order (the LDS itself executes one wave's operations in issue order -- verified in r02 with and without an
// lgkmcnt(0) between the writes and the reads).
__device__ __forceinline__ void pp_lds_fence() { asm volatile("" ::: "memory"); }

// 16-byte buffer store followed by the wait states hipcc does not insert: with an SGPR soffset LLVM's hazard recognizer assumes
// "store of more than 64 bits -> VALU overwrite of its data registers" cannot happen, but on gfx950 the very next VALU write DID
// corrupt the stored dwords (r02: 0.7 % of the f32-epilogue outputs, 1 % of the GELU outputs, always the same lanes).
template <int AUX = 0>      // AUX 16 = sc1 (write-through): the tile is handed to another workgroup inside the launch
__device__ __forceinline__ void pp_store_b128(u32x4 d, __amdgpu_buffer_rsrc_t ro, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b128(d, ro, voff, soff, AUX);
    __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 1" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
}

// ---- full-tile epilogue through a wave-private 4 KiB LDS patch: whole-row stores, an EXACT number of vector-memory instructions.
//   * the accumulators hold one ROW per lane: a store straight from them touches 16 different 128-byte lines per instruction and
//     the texture-address path serialises on lines (the ring kernels lost 20-27 % to it when they first moved to 16-wide tiles).
//     So each 16-row block goes through the patch: written in the MFMA layout, read back with 8 lanes per 128-byte row, stored as
//     whole lines (8 lines per instruction).  Patch rows are 128 B; 16-byte slots are XOR-ed with (row & 7): conflict-free reads,
//     <= 2-way writes.
//   * every store is one buffer_store_dwordx4 issued unconditionally (16 * NB / 4 for 16-bit outputs, 8 * NB for f32), so a
//     persistent kernel can skip over them with a counted vmcnt in the next tile's K loop instead of draining them.
// The wave's block is (NB * 32) rows x 64 columns: acc[2 NB][4].  bq[u] = bias of columns 16 u + 4 g4 .. + 3.
// voff = this lane's byte offset in row layout (row lane >> 3 of the wave's block, 16-byte piece lane & 7), soff = tile origin,
// soff8 = 8 rows, all in bytes of the output type; ro = buffer resource of the output matrix.
template <typename T, int EPI, int NB, int AUX = 0, typename ACC = AccPlain>
__device__ __forceinline__ void epilogue16_staged(f32x4 (&acc)[2 * NB][4], const f32x4 (&bq)[4], __amdgpu_buffer_rsrc_t ro, char *patch, int voff, int soff, int soff8, int lane, int hilo_soff = 0, int idx0 = 0) {
    // The 4 KiB patch is used as TWO halves of 16 rows x 128 B (r03): block b + 1 is converted and written into the other half between the
    // issue of block b's read-back and its stores, so the LDS write -> read round trip and the stores' issue time hide under the next
    // block's VALU work instead of serialising once per block.  (The LDS executes a wave's operations in issue order; the compiler-level
    // fences keep the program order of accesses it cannot prove distinct.)
    const int l15 = lane & 15, g4 = lane >> 4;
    const int rd_off = (lane >> 3) * 128 + (((lane & 7) ^ ((lane >> 3) & 7)) * 16);         // row layout: row (lane>>3) + 8t, 16-byte piece lane&7
    const int wr_row = l15 * 128, x16 = (l15 & 7) * 16;                                    // MFMA layout: this lane's row of a 16-row block
    pp_lds_fence();
    if constexpr (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_HILO) {
        typedef typename Pair<T>::v2 v2;
        // EPI_BIAS_HILO stores every 16-row block twice: pass q = 2 b + plane, plane 0 = hi at the tile's place, plane 1 = lo behind
        // `hilo_soff` bytes (2 x the stores: 32 per NB = 4 tile, what pp_epi_stores counts)
        constexpr int PL = EPI == EPI_BIAS_HILO ? 2 : 1;
        auto convert_write = [&](int q) {            // 16-row block q / PL = accumulator tile of the wave, into half q & 1 of the patch
            const int b = q / PL, plane = q % PL;
            char *pb = patch + (q & 1) * 2048 + wr_row;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const f32x4 v = ACC::get(acc[b][u], b * 8 + u + idx0) + bq[u];
                v2 p0, p1;
                if constexpr (EPI == EPI_BIAS_GELU) { p0 = gelu_out_pair<T>(v[0], v[1]); p1 = gelu_out_pair<T>(v[2], v[3]); }
                else { p0 = round_pair<T>(v[0], v[1]); p1 = round_pair<T>(v[2], v[3]); }
                if (PL == 2 && plane == 1) { p0 = hilo_lo_pair<T>(v[0], v[1], p0); p1 = hilo_lo_pair<T>(v[2], v[3], p1); }
                // columns u * 16 + 4 g4 .. + 3 -> bytes u * 32 + 8 g4 of the 128-byte patch row: 16-byte slot 2u + (g4 >> 1), half g4 & 1
                *(u32x2 *)(pb + (((2 * u + (g4 >> 1)) * 16) ^ x16) + (g4 & 1) * 8) = u32x2{__builtin_bit_cast(unsigned, p0), __builtin_bit_cast(unsigned, p1)};
            }
            pp_lds_fence();
        };
        convert_write(0);
#pragma unroll
        for (int q = 0; q < 2 * NB * PL; ++q) {
            const int b = q / PL, plane = q % PL;
            u32x4 d[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) d[t] = *(const u32x4 *)(patch + (q & 1) * 2048 + t * 1024 + rd_off);
            pp_lds_fence();
            if (q + 1 < 2 * NB * PL) convert_write(q + 1);
#pragma unroll
            for (int t = 0; t < 2; ++t) pp_store_b128<AUX>(d[t], ro, voff, soff + (b * 2 + t) * soff8 + (plane ? hilo_soff : 0));
        }
    } else {       // f32 outputs: one 16 x 32 block (2 KiB) per pass, pass c = (16-row block c >> 1, 32-column half c & 1)
        u32x4 res[2][2];                            // residual rows of the current and the next pass (loads run one pass ahead)
        auto load_res = [&](int c, u32x4 (&dst)[2]) {
            const int b = c >> 1, j = c & 1;
#pragma unroll
            for (int t = 0; t < 2; ++t) dst[t] = __builtin_amdgcn_raw_buffer_load_b128(ro, voff + j * 128, soff + (b * 2 + t) * soff8, 0);
        };
        auto write_pass = [&](int c) {
            const int b = c >> 1, j = c & 1;
            char *pb = patch + (c & 1) * 2048 + wr_row;
#pragma unroll
            for (int uu = 0; uu < 2; ++uu)          // columns (2j + uu) * 16 + 4 g4 of the wave = (uu * 16 + 4 g4) of this 32-column half: slot 4 uu + g4
                *(f32x4 *)(pb + (((4 * uu + g4) * 16) ^ x16)) = ACC::get(acc[b][2 * j + uu], b * 8 + 2 * j + uu + idx0) + bq[2 * j + uu];
            pp_lds_fence();
        };
        if constexpr (EPI == EPI_BIAS_RESID) load_res(0, res[0]);
        write_pass(0);
#pragma unroll
        for (int c = 0; c < 4 * NB; ++c) {
            const int b = c >> 1, j = c & 1;
            if constexpr (EPI == EPI_BIAS_RESID) { if (c + 1 < 4 * NB) load_res(c + 1, res[(c + 1) & 1]); }
            f32x4 d[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) d[t] = *(const f32x4 *)(patch + (c & 1) * 2048 + t * 1024 + rd_off);
            pp_lds_fence();
            if (c + 1 < 4 * NB) write_pass(c + 1);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if constexpr (EPI == EPI_BIAS_RESID) d[t] = d[t] + __builtin_bit_cast(f32x4, res[c & 1][t]);     // (acc + bias) + x, the reference's order (vit.cpp:868-873)
                pp_store_b128<AUX>(__builtin_bit_cast(u32x4, d[t]), ro, voff + j * 128, soff + (b * 2 + t) * soff8);
            }
        }
    }
}

// Whole-workgroup-tile wrapper for the kernels that stage through LDS they own after their K loop: the staged epilogue when the
// tile is full and its byte offsets fit the 32-bit buffer addressing, epilogue16 otherwise (edge tiles, the patch embedding).
// wave_row0 / wave_col0 = origin of the wave's (NB * 32) x 64 block inside the workgroup tile at (m0, n0).
template <typename T, int EPI, int NB>
__device__ __forceinline__ void epilogue16_tile(const GemmArgs &g, f32x4 (&acc)[2 * NB][4], bool full, int m0, int n0, int wave_row0, int wave_col0, char *patch, int lane) {
    const int l15 = lane & 15, g4 = lane >> 4;
    if constexpr (EPI != EPI_PATCH) {
        constexpr int esz = (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_HILO) ? 2 : 4;
        if (full && (size_t)g.M * g.ldo * esz + (EPI == EPI_BIAS_HILO ? (size_t)g.hilo_off * esz : 0) < 0xf0000000u && (g.ldo & 3) == 0) {
            f32x4 bq[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) bq[u] = g.bias ? *(const f32x4 *)(g.bias + n0 + wave_col0 + u * 16 + g4 * 4) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(g.out, 0, (int)0xffffffffu, 0x00020000);
            const int voff = ((wave_row0 + (lane >> 3)) * g.ldo + wave_col0) * esz + (lane & 7) * 16;
            epilogue16_staged<T, EPI, NB>(acc, bq, ro, patch, voff, __builtin_amdgcn_readfirstlane((m0 * g.ldo + n0) * esz), 8 * g.ldo * esz, lane, (int)(g.hilo_off * esz));
            return;
        }
    }
    if (full) epilogue16<T, EPI, 2 * NB, 4, true>(g, acc, m0 + wave_row0 + l15, n0 + wave_col0 + 4 * g4);
    else epilogue16<T, EPI, 2 * NB, 4, false>(g, acc, m0 + wave_row0 + l15, n0 + wave_col0 + 4 * g4);
}

}  // namespace vitx

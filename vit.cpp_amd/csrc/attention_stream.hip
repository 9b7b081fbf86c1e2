// attention_stream.hip -- streaming two-pass attention for gfx950 (vit.cpp:826-866), any token count, head dim 64.
//
// Two builds of one kernel:
//   * fast  (QT = 2, PREC = false; f16 or bf16 operands): built as the r03 verdict's long-sequence kernel (577 tokens of ViT-L/16-384) --
//     v_mfma_f32_16x16x32 (the GEMMs' instruction), a wave owns TWO 16-query tiles so every K / V^T fragment read from LDS feeds two products
//     (one fragment per product is exactly the LDS peak: 1 KiB per 16-cycle MFMA per SIMD = 256 B/clk/CU), a three-slot ring of 64-key chunks
//     filled by LDS-DMA two chunks ahead with counted vmcnt and one raw barrier per chunk, every LDS read inline asm with counted lgkmcnt, row
//     reductions by permlane swaps, 16-byte output stores; 8 waves x 256 VGPRs (r04: 16 x 128, measured equal).  MEASURED SLOWER than attention_flow_kernel at 577 tokens
//     (250 vs 212 us, profiles/r04/attention_577_ablation.txt): it is kernel id 5 (tests, lab) and NOT what the forward runs in bf16.
//   * precise (QT = 1, PREC = true; f16 only): the F16 PARITY MODE at every token count.  The reference multiplies f32 q, k, v
//     (ggml_mul_mat on f32 views, vit.cpp:848,858); r03 rounded them to fp16 for the MFMAs -- the one known semantic deviation of that
//     mode.  Here q, k, v arrive as TWO fp16 planes from the QKV GEMM (EPI_BIAS_HILO: hi = round(x), lo = round((x - hi) * 2048)) and
//     every product is three MFMAs, hi.hi + (hi.lo + lo.hi) / 2048 (the dropped lo.lo term is 2^-22 relative): f32-grade scores; the
//     probabilities are the fp16 exp-table values (exact in fp16, as ggml's LUT emits them), so P.V needs only V split: two MFMAs.
//     193..224 tokens (every headline model) take attention_precise_kernel below (persistent; score tiles in registers between ONE K stream
//     and the V stream); every other token count this kernel's two-pass build, 8 waves x 256 VGPRs.
// Every build claims the whole register file of its SIMDs (AS_CLAIM below): a foreign wave beside this MFMA stream computed wrong DPP sums.
// Both: pass 1 streams K for the row maxima of the raw scores, pass 2 streams K and V: e = AttnExp<T>(s, max) (F16: ggml_soft_max's
// table semantics), row sum of the rounded numerators, O^T = V^T . P^T, O / sum rounded once to the operand type.
// Keys past N inside the last chunk read the next image's rows (finite; masked to -inf, probability exactly 0) or the zeros a buffer
// load returns out of range.
#include <type_traits>

#include "device_common.h"
#include "kernels.h"

namespace vitx {

namespace as {
#ifndef AS_W
#define AS_W 8          // waves per workgroup of the fast build (r05: 8 x 256 registers -- at 16 x 128 it spilled 25 registers, and a spill is a vector-memory operation inside its counted vmcnt schedule)
#endif
#ifndef AS_OCC
#define AS_OCC 4        // waves per SIMD the fast build is compiled for (the pipelined step, AS_PIPE, needs 194 VGPRs: AS_OCC 2)
#endif
#ifndef AS_NSLOT
#define AS_NSLOT 3      // ring slots of the fast build
#endif
#ifndef AS_FLAGS
#define AS_FLAGS 0      // ablation builds (tools/scratch only; garbage results): 1 no exp arithmetic, 2 no V reads / PV products, 4 no K reads, 8 no QK^T products, 16 no DMA / barriers after the prologue, 32 no pass 1
#endif
constexpr int FL = AS_FLAGS;
#ifndef AS_PIPE
#define AS_PIPE 0       // fast build: 1 = software-pipelined steps (PV of step s - 1 beside the exp arithmetic of step s), 0 = the plain per-step schedule
#endif
constexpr int CK = 64;               // keys per chunk
constexpr int KB = CK * 128;         // bytes of one 64-row plane image (K or V rows of 64 dims x 2 B)
#ifndef AS_PNSLOT
#define AS_PNSLOT 3
#endif
#ifndef AS_AUX
#define AS_AUX 0          // cache-policy bits of the K / V DMA and the Q loads (lab: 17 = sc0 sc1)
#endif
#ifndef AS_CLAIM
#define AS_CLAIM 1
#endif
#ifndef AP_DROP
#define AP_DROP 0         // numerics experiment (r04 verdict item 3; profiles/r05/precise_attention_dropped_terms.txt, r06: profiles/r06/precise_attention_planes.txt): 1 = no k_lo . q_hi products, 2 = no P . v_lo products,
                          // 4 = no k_hi . q_lo products.  r06: the lo plane of an operand whose products are dropped is not FETCHED either (attention_precise_kernel: a zero-length buffer descriptor returns zeros and fetches nothing)
#endif
#ifndef AP_K2
#define AP_K2 1           // K fragments of both 16-key tiles of a 32-key step in flight (16 more registers, still no spill): F16 forward 11.02 -> 10.94 ms, same bits (profiles/r05/ab_precise_k2.txt)
#endif
#ifndef AP_AUX
#define AP_AUX 2          // cache-policy bits of attention_precise_kernel's Q / K / V LDS-DMA: nt -- the two QKV planes are read once (F16 forward 11.20 -> 11.07 ms, profiles/r05/ab_cache_policy2.txt)
#endif
#ifndef AS_PARANOID
#define AS_PARANOID 0
#endif
template <bool PREC> constexpr int nslot() { return PREC ? AS_PNSLOT : AS_NSLOT; }
template <bool PREC> constexpr int nwaves() { return PREC ? 8 : AS_W; }       // the two-pass precise build needs 150 registers: 8 waves
template <bool PREC> constexpr int slot_bytes() { return (PREC ? 4 : 2) * KB; }      // [K hi | K lo | V hi | V lo] or [K | V]
template <int I, int N, typename F> __device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_vm_rt(int n) {       // n = DMA instructions allowed to stay in flight (wave-uniform)
    switch (n) {
    case 0: wait_vm<0>(); break; case 1: wait_vm<1>(); break; case 2: wait_vm<2>(); break; case 3: wait_vm<3>(); break; case 4: wait_vm<4>(); break;
    case 5: wait_vm<5>(); break; case 6: wait_vm<6>(); break; case 7: wait_vm<7>(); break; case 8: wait_vm<8>(); break; case 9: wait_vm<9>(); break;
    case 10: wait_vm<10>(); break; case 11: wait_vm<11>(); break; case 12: wait_vm<12>(); break; case 13: wait_vm<13>(); break; case 16: wait_vm<16>(); break;
    default: wait_vm<0>(); break;
    }
}
typedef int i4 __attribute__((ext_vector_type(4)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef short s8 __attribute__((ext_vector_type(8)));
template <int OFF> __device__ __forceinline__ void ds_read_b128(i4 &dst, unsigned addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF)); }
template <int OFF> __device__ __forceinline__ void ds_read_tr(s4 &dst, unsigned addr) { asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF)); }
// counted LDS waits that OWN the registers the reads deliver into (the products that consume them cannot be scheduled above the wait);
// free functions: clang does not capture a variable a lambda names only in an asm operand
template <int CNT> __device__ __forceinline__ void wait_lgkm4(i4 &a, i4 &b, i4 &c, i4 &d) { asm volatile("s_waitcnt lgkmcnt(%[n])" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : [n] "n"(CNT)); }
template <int CNT> __device__ __forceinline__ void wait_lgkm8(i4 &a, i4 &b, i4 &c, i4 &d, i4 &e, i4 &f, i4 &g, i4 &h) {
    asm volatile("s_waitcnt lgkmcnt(%[n])" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : [n] "n"(CNT));
}
template <int CNT> __device__ __forceinline__ void wait_lgkm8(s4 &a, s4 &b, s4 &c, s4 &d, s4 &e, s4 &f, s4 &g, s4 &h) {
    asm volatile("s_waitcnt lgkmcnt(%[n])" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : [n] "n"(CNT));
}
}  // namespace as

template <typename T, int QT, bool PREC, int W, int NSLOT>
__global__ __launch_bounds__(W * 64, W == 16 ? 4 : 2) void attention_stream_kernel(const T *__restrict__ qkv, T *__restrict__ out, int N, int D, int H, int items, int qblocks, int n_img, long lo_off) {
    using namespace as;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // a plane image is 512 pieces of 16 bytes: moved by the first DT = min(NT, 512) threads, OPS pieces each (waves past them only compute)
    constexpr int PL = PREC ? 2 : 1, SLOT = slot_bytes<PREC>(), VBASE = PL * KB, NT = W * 64, DT = NT < 512 ? NT : 512, OPS = 512 / DT, AHEAD = NSLOT - 1;
    static_assert(512 % DT == 0 && NSLOT >= 2, "a plane image is 512 pieces of 16 bytes");
    typedef typename Elem<T>::v8 v8;
    typedef typename Pair<T>::v2 v2;
#if AS_CLAIM
    // Claim the whole register file of every SIMD the workgroup occupies, as the GEMM and the persistent attention kernels do by their size:
    // 8 waves x 256 registers, or 16 waves x 128.  r04: with registers to spare, waves of ANOTHER kernel (the other sub-batch stream's
    // LayerNorm) were placed on the same SIMDs and their cross-lane sums came out wrong in ~1 % of the rows (profiles/r04/coresidency_layernorm.txt)
    // -- an interaction between this kernel's instruction stream and a co-resident wave's DPP / permute traffic that nothing in the ISA
    // documents and that the kernels which fill their CUs never showed.  A full register file admits no foreign wave.
    if constexpr (W == 8) asm volatile("v_mov_b32 v255, 0" ::: "v255");
    else { static_assert(W == 16, "8 waves x 256 or 16 waves x 128 registers"); asm volatile("v_mov_b32 v127, 0" ::: "v127"); }
#endif
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g4 = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // blockIdx -> (item, query block): blocks equal mod 8 run on one XCD; an item's query blocks are consecutive there (its K / V stay in that L2)
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int item = (jb / qblocks) * 8 + xcd, qb = jb % qblocks;
    if (item >= items) return;
    const int b = item / H, h = item - b * H;
    const T *base = qkv + (size_t)b * N * 3 * D + h * 64;
    const int row_bytes = 3 * D * 2;
    // tasks = 16 QT-query tiles of this item, dealt to the item's query blocks in contiguous, balanced runs; wave w takes task first + w
    const int tasks = (N + 16 * QT - 1) / (16 * QT), per = tasks / qblocks, extra = tasks % qblocks;
    const int first = qb * per + min(qb, extra), mine = per + (qb < extra ? 1 : 0);
    const bool active = wave < mine;                     // a wave without a task only moves data and keeps the barriers
    const int q0 = (first + wave) * 16 * QT;
    const int nch = (N + CK - 1) / CK, nstage = 2 * nch;

    // ---- LDS-DMA: physical 16-B piece tid of a plane image <-> (key row, 16-B piece) of the K / V column block of this head
    const unsigned remaining = (unsigned)min((size_t)0xf0000000u, ((size_t)(n_img - b) * N * 3 * D - h * 64) * 2);
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)remaining, 0x00020000);
    __amdgpu_buffer_rsrc_t rsrc_lo = __builtin_amdgcn_make_buffer_rsrc((void *)(base + (PREC ? lo_off : 0)), 0, (int)remaining, 0x00020000);
    int koff[2], voff[2];           // (fixed bound, OPS <= 2: hipcc 7.2 silently drops the host-side instantiation of a kernel whose lambda captures an array sized by a constexpr LOCAL that depends on a template parameter)
    static_assert(OPS <= 2, "at least 4 waves");
#pragma unroll
    for (int r = 0; r < OPS; ++r) {
        const int piece = r * DT + (tid & (DT - 1));
        int rr, sl; swz_inv(piece, rr, sl);                                        // K: swizzled row image (swz_byte), permutation on the source side
        koff[r] = rr * row_bytes + D * 2 + sl * 16;
        const int vr = piece >> 3, vs = (piece & 7) ^ (((vr >> 1) & 3) << 1);      // V: row-major, 32-byte chunks XOR-ed with (row >> 1) & 3
        voff[r] = vr * row_bytes + 2 * D * 2 + vs * 16;
    }
    const bool dma_wave = wave * 64 < DT;               // wave-uniform
    auto stage = [&](int i) {            // stage i: pass-1 chunk i (K only) for i < nch, pass-2 chunk i - nch (K and V) after that
        if (!dma_wave) return;
        const bool with_v = i >= nch;
        const int c = with_v ? i - nch : i;
        char *dst = smem + (i % NSLOT) * SLOT + wave * 1024;
        const int so = c * CK * row_bytes;
#pragma unroll
        for (int r = 0; r < OPS; ++r) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LPTR(dst + r * DT * 16), 16, koff[r], so, 0, AS_AUX);
            if (PREC) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_lo, LPTR(dst + r * DT * 16 + KB), 16, koff[r], so, 0, AS_AUX);
        }
        if (with_v) {
#pragma unroll
            for (int r = 0; r < OPS; ++r) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LPTR(dst + r * DT * 16 + VBASE), 16, voff[r], so, 0, AS_AUX);
                if (PREC) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_lo, LPTR(dst + r * DT * 16 + VBASE + KB), 16, voff[r], so, 0, AS_AUX);
            }
        }
    };
    auto stage_ops = [&](int i) { return (i >= nstage || !dma_wave) ? 0 : (i >= nch ? 2 * PL * OPS : PL * OPS); };
    // DMA instructions of stages i + 2 .. i + AHEAD: what may stay in flight when stage i + 1 must have landed
    auto ops_after = [&](int i) { int n = 0; for (int a = 2; a <= AHEAD; ++a) n += stage_ops(i + a); return n; };

    // ---- Q fragments (B operand of S^T = K . Q^T): lane (l15 = query of the tile, g4) holds dims k2 * 32 + g4 * 8 .. + 7
    v8 qh[QT][2], ql[QT][2];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int qrow = min(q0 + qt * 16 + l15, N - 1);
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            const int qo = (qrow * 3 * D + k2 * 32 + g4 * 8) * 2;
            qh[qt][k2] = __builtin_bit_cast(v8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, qo, 0, AS_AUX));
            ql[qt][k2] = PREC ? __builtin_bit_cast(v8, __builtin_amdgcn_raw_buffer_load_b128(rsrc_lo, qo, 0, AS_AUX)) : qh[qt][k2];
        }
    }
#pragma unroll
    for (int a = 0; a < AHEAD; ++a) if (a < nstage) stage(a);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) { asm volatile("" : "+v"(qh[qt][k2])); if (PREC) asm volatile("" : "+v"(ql[qt][k2])); }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    // ---- fragment addresses inside a slot: K tile t (16 keys) = parity (t & 1) base + (t >> 1) * 4096; V step ks (32 keys) adds ks * 4096, its second 16 keys 2048
    const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)((__attribute__((address_space(3))) char *)smem);
    int krd[2][2], vrd[4];
#pragma unroll
    for (int pz = 0; pz < 2; ++pz)
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) krd[pz][k2] = swz_byte(pz * 16 + l15, k2 * 4 + g4);
    {
        const int r = 4 * g4 + (l15 >> 2), x = (r >> 1) & 3;       // this lane's V row within a 16-key group and its chunk swizzle
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) vrd[dt] = VBASE + r * 128 + ((dt ^ x) << 5) + (l15 & 3) * 8;
    }

    // K fragments of one 32-key step: [16-key tile j][k2] (+ the lo plane)
    i4 kf[2][2], kl[2][2];
    auto read_k = [&](unsigned sb, auto ks_) {
        constexpr int ks = decltype(ks_)::value;
        if constexpr ((FL & 4) != 0) {
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) { kf[0][k2] = i4{(int)sb, ks, k2, 1}; kf[1][k2] = i4{ks, (int)sb, 2, k2}; }
            return;
        }
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            ds_read_b128<ks * 4096>(kf[0][k2], sb + krd[0][k2]); ds_read_b128<ks * 4096>(kf[1][k2], sb + krd[1][k2]);
            if constexpr (PREC) { ds_read_b128<ks * 4096 + KB>(kl[0][k2], sb + krd[0][k2]); ds_read_b128<ks * 4096 + KB>(kl[1][k2], sb + krd[1][k2]); }
        }
    };
    auto wait_k = [&](auto cnt_) {       // the K reads have landed; cnt = LDS reads issued after them
        constexpr int cnt = decltype(cnt_)::value;
        if constexpr ((FL & 4) != 0) return;
        if constexpr (PREC) wait_lgkm8<cnt>(kf[0][0], kf[0][1], kf[1][0], kf[1][1], kl[0][0], kl[0][1], kl[1][0], kl[1][1]);
        else wait_lgkm4<cnt>(kf[0][0], kf[0][1], kf[1][0], kf[1][1]);
    };
    // S^T tile: rows = 16 keys, cols = the 16 queries of tile qt; lane (l15, g4) gets keys 4 g4 .. + 3 of query l15
    auto score = [&](int j, int qt) -> f32x4 {
        const v8 k0 = __builtin_bit_cast(v8, kf[j][0]), k1 = __builtin_bit_cast(v8, kf[j][1]);
        f32x4 a = {0.0f, 0.0f, 0.0f, 0.0f};
        if constexpr ((FL & 8) != 0) return f32x4{(float)k0[0], (float)k1[1], (float)k0[2] + (float)qh[qt][0][0], (float)k1[3]};
        a = Elem<T>::mfma16(k0, qh[qt][0], a);
        a = Elem<T>::mfma16(k1, qh[qt][1], a);
        if constexpr (PREC) {
            const v8 l0 = __builtin_bit_cast(v8, kl[j][0]), l1 = __builtin_bit_cast(v8, kl[j][1]);
            f32x4 c = {0.0f, 0.0f, 0.0f, 0.0f};
            c = Elem<T>::mfma16(k0, ql[qt][0], c);
            c = Elem<T>::mfma16(k1, ql[qt][1], c);
            c = Elem<T>::mfma16(l0, qh[qt][0], c);
            c = Elem<T>::mfma16(l1, qh[qt][1], c);
#pragma unroll
            for (int r = 0; r < 4; ++r) a[r] = __builtin_fmaf(c[r], kHiLoInv, a[r]);
        }
        return a;
    };
    auto mask = [&](f32x4 &s, int key0) {            // key0 = first key of the tile
#pragma unroll
        for (int r = 0; r < 4; ++r) if (key0 + 4 * g4 + r >= N) s[r] = -INFINITY;
    };

    // =================================================== pass 1: row maxima of the raw scores
    float mx[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) mx[qt] = -INFINITY;
    auto p1_step = [&](unsigned sb, int key0, auto ks_, auto masked_) {
        constexpr int ks = decltype(ks_)::value;
        constexpr bool MASK = decltype(masked_)::value;
        read_k(sb, ks_);
        wait_k(std::integral_constant<int, 0>{});
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                f32x4 s = score(j, qt);
                if (MASK) mask(s, key0 + ks * 32 + j * 16);
                mx[qt] = fmaxf(fmaxf(mx[qt], s[0]), s[1]);       // v_max3_f32
                mx[qt] = fmaxf(fmaxf(mx[qt], s[2]), s[3]);
            }
    };
    typedef std::integral_constant<int, 0> I0; typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<int, 2> I2; typedef std::integral_constant<int, 4> I4; typedef std::integral_constant<int, 8> I8;
    typedef std::true_type TT; typedef std::false_type FF;
    constexpr bool PIPE = !PREC && AS_PIPE != 0;
    // fast build: the 8 products of a 32-key step (2 key tiles x QT query tiles x 2 k-steps) in k-step-major order -- four (2 QT) independent
    // accumulator chains, so the second product of a chain is issued four products (64 cycles) after the first instead of right behind it
    auto qk8 = [&](f32x4 (&sc)[2][QT]) {
        v8 k[2][2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) k[j][k2] = __builtin_bit_cast(v8, kf[j][k2]);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                if constexpr ((FL & 8) != 0) sc[j][qt] = f32x4{(float)k[j][0][0], (float)k[j][1][1], (float)k[j][0][2] + (float)qh[qt][0][0], (float)k[j][1][3]};
                else sc[j][qt] = Elem<T>::mfma16(k[j][0], qh[qt][0], f32x4{0.0f, 0.0f, 0.0f, 0.0f});
            }
        if constexpr ((FL & 8) == 0) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) sc[j][qt] = Elem<T>::mfma16(k[j][1], qh[qt][1], sc[j][qt]);
        }
    };
    auto p1_pipe = [&](unsigned sb, int key0, auto masked_, bool two) {      // one chunk of pass 1: the K fragments of step 1 are requested behind the products of step 0
        constexpr bool MASK = decltype(masked_)::value;
        f32x4 sc[2][QT];
        read_k(sb, I0{}); wait_k(I0{});
        qk8(sc);
        if (two) read_k(sb, I1{});
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                if (MASK) mask(sc[j][qt], key0 + j * 16);
                mx[qt] = fmaxf(fmaxf(mx[qt], sc[j][qt][0]), sc[j][qt][1]); mx[qt] = fmaxf(fmaxf(mx[qt], sc[j][qt][2]), sc[j][qt][3]);
            }
        if (two) {
            wait_k(I0{});
            qk8(sc);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    if (MASK) mask(sc[j][qt], key0 + 32 + j * 16);
                    mx[qt] = fmaxf(fmaxf(mx[qt], sc[j][qt][0]), sc[j][qt][1]); mx[qt] = fmaxf(fmaxf(mx[qt], sc[j][qt][2]), sc[j][qt][3]);
                }
        }
    };
    for (int c = 0; c < nch; ++c) {
        const int i = c;
        if ((FL & 32) != 0) break;
        if (AS_PARANOID) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
        if (i + AHEAD < nstage && !(FL & 16)) stage(i + AHEAD);
        __builtin_amdgcn_sched_barrier(0);
        if (active) {
            const unsigned sb = lds0 + (unsigned)((i % NSLOT) * SLOT);
            const int key0 = c * CK;
            if constexpr (PIPE) {
                if (c + 1 < nch) p1_pipe(sb, key0, FF{}, true); else p1_pipe(sb, key0, TT{}, key0 + 32 < N);
            } else {
                if (c + 1 < nch) { p1_step(sb, key0, I0{}, FF{}); p1_step(sb, key0, I1{}, FF{}); }
                else { p1_step(sb, key0, I0{}, TT{}); if (key0 + 32 < N) p1_step(sb, key0, I1{}, TT{}); }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (FL & 16) continue;
        wait_vm_rt(ops_after(i));                       // stage i + 1 has landed; the pieces of the stages after it may stay in flight
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    float nmx[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) nmx[qt] = -AttnExp<T>::kScale * rows4_max(mx[qt]);

    // =================================================== pass 2: numerators, row sums, O^T = V^T . P^T
    f32x4 o[QT][4], oc[QT][4];
    float sum[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        sum[qt] = 0.0f;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { o[qt][dt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; oc[qt][dt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
    }
    s4 vf[4][2], vl[4][2];
    // V^T fragments of 16-dim tiles DT0 .. DT0 + NDT - 1 of one 32-key step (hi plane, and lo for PREC)
    auto read_v = [&](unsigned sb, auto ks_, auto dt0_, auto ndt_) {
        constexpr int ks = decltype(ks_)::value, DT0 = decltype(dt0_)::value, NDT = decltype(ndt_)::value;
        if constexpr ((FL & 2) != 0) return;
        static_for<DT0, DT0 + NDT>([&](auto dt_) {
            constexpr int dt = decltype(dt_)::value;
            ds_read_tr<ks * 4096>(vf[dt][0], sb + vrd[dt]); ds_read_tr<ks * 4096 + 2048>(vf[dt][1], sb + vrd[dt]);
            if constexpr (PREC) { ds_read_tr<ks * 4096 + KB>(vl[dt][0], sb + vrd[dt]); ds_read_tr<ks * 4096 + 2048 + KB>(vl[dt][1], sb + vrd[dt]); }
        });
    };
    auto pv = [&](const v8 (&p)[QT], auto dt0_, auto ndt_) {
        constexpr int DT0 = decltype(dt0_)::value, NDT = decltype(ndt_)::value;
        if constexpr ((FL & 2) != 0) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) o[qt][DT0][0] += (float)p[qt][0] + (float)p[qt][5];
            return;
        }
        static_for<DT0, DT0 + NDT>([&](auto dt_) {
            constexpr int dt = decltype(dt_)::value;
            const s8 both = __builtin_shufflevector(vf[dt][0], vf[dt][1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) o[qt][dt] = Elem<T>::mfma16(__builtin_bit_cast(v8, both), p[qt], o[qt][dt]);
            if constexpr (PREC) {
                const s8 bl = __builtin_shufflevector(vl[dt][0], vl[dt][1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) oc[qt][dt] = Elem<T>::mfma16(__builtin_bit_cast(v8, bl), p[qt], oc[qt][dt]);
            }
        });
    };
    auto p2_step = [&](unsigned sb, int key0, auto ks_, auto masked_) {
        constexpr bool MASK = decltype(masked_)::value;
        constexpr int ks = decltype(ks_)::value;
        read_k(sb, ks_);
        if constexpr (!PREC) { read_v(sb, ks_, I0{}, I4{}); if constexpr ((FL & 2) != 0) wait_k(I0{}); else wait_k(I8{}); }          // 4 + 8 LDS reads in flight (the counter holds 15)
        else wait_k(I0{});
        // numerators per AttnExp<T>, one 16-key tile at a time (its four scores per query die as soon as they are exponentiated); row sum of the
        // ROUNDED values (they are what the PV product sees); k-slot j of lane group g4 = key 4 g4 + j of the first, 16 + 4 g4 + (j - 4) of
        // the second 16-key tile of the step -- the order the transposed V reads deliver
        v2 e[QT][4];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                f32x4 sc = score(j, qt);
                if (MASK) mask(sc, key0 + ks * 32 + j * 16);
                if constexpr ((FL & 1) != 0) { e[qt][2 * j] = round_pair<T>(sc[0] + nmx[qt], sc[1]); e[qt][2 * j + 1] = round_pair<T>(sc[2], sc[3]); }
                else { e[qt][2 * j] = AttnExp<T>::pair(sc[0], sc[1], nmx[qt]); e[qt][2 * j + 1] = AttnExp<T>::pair(sc[2], sc[3], nmx[qt]); }
                sum[qt] = Pair<T>::sum2(e[qt][2 * j + 1], Pair<T>::sum2(e[qt][2 * j], sum[qt]));
            }
            if (PREC && j == 0) read_v(sb, ks_, I0{}, I2{});                           // 8 reads: dims 0..31, both planes (the K fragments of tile 1 are in registers already)
        }
        v8 p[QT];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) p[qt] = v8{e[qt][0][0], e[qt][0][1], e[qt][1][0], e[qt][1][1], e[qt][2][0], e[qt][2][1], e[qt][3][0], e[qt][3][1]};
        if constexpr (!PREC) {
            if constexpr ((FL & 2) == 0) wait_lgkm8<0>(vf[0][0], vf[0][1], vf[1][0], vf[1][1], vf[2][0], vf[2][1], vf[3][0], vf[3][1]);
            pv(p, I0{}, I4{});
        } else {
            wait_lgkm8<0>(vf[0][0], vf[0][1], vf[1][0], vf[1][1], vl[0][0], vl[0][1], vl[1][0], vl[1][1]);
            pv(p, I0{}, I2{});
            read_v(sb, ks_, I2{}, I2{});
            wait_lgkm8<0>(vf[2][0], vf[2][1], vf[3][0], vf[3][1], vl[2][0], vl[2][1], vl[3][0], vl[3][1]);
            pv(p, I2{}, I2{});
        }
    };
    // fast build, software-pipelined (r04: the plain per-step schedule ran every phase at its full serial cost -- the waves of a SIMD go through
    // reads -> QK^T -> exp -> PV in lock step, one barrier per chunk re-aligns them -- profiles/r04/attention_577_ablation.txt): the PV products
    // of step s - 1 (operands: last step's probabilities and its V^T fragments, both in registers) are issued BESIDE the exp arithmetic of
    // step s, so matrix pipe and VALU of the same wave run together; the K fragments of the chunk's second step are requested right behind the
    // products of its first, the V^T fragments of step s behind the PV products that free their registers.
    v8 p_prev[QT];
    if constexpr (PIPE) {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) p_prev[qt] = __builtin_bit_cast(v8, i4{0, 0, 0, 0});
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { vf[dt][0] = s4{0, 0, 0, 0}; vf[dt][1] = s4{0, 0, 0, 0}; }       // zero probabilities x zero V: the first PV adds nothing
    }
    auto v_landed = [&]() { if constexpr ((FL & 2) == 0) wait_lgkm8<0>(vf[0][0], vf[0][1], vf[1][0], vf[1][1], vf[2][0], vf[2][1], vf[3][0], vf[3][1]); };
    auto p2_pipe = [&](unsigned sb, int key0, auto ks_, auto masked_, bool more) {
        constexpr bool MASK = decltype(masked_)::value;
        constexpr int ks = decltype(ks_)::value;
        if constexpr (ks == 0) { read_k(sb, I0{}); wait_k(I0{}); }
        else { if constexpr ((FL & 2) != 0) wait_k(I0{}); else wait_k(I8{}); }       // K of step 1 was requested BEFORE the 8 V^T reads of step 0
        __builtin_amdgcn_sched_barrier(0);
        f32x4 sc[2][QT];
        qk8(sc);
        if constexpr (ks == 0) { if (more) read_k(sb, I1{}); }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ks == 1) v_landed();                                           // V^T of step 0 (younger than the K reads waited for above)
        pv(p_prev, I0{}, I4{});                                                      // step s - 1: 8 products ...
        v2 e[QT][4];                                                                 // ... beside this step's numerators
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                if (MASK) mask(sc[j][qt], key0 + ks * 32 + j * 16);
                if constexpr ((FL & 1) != 0) { e[qt][2 * j] = round_pair<T>(sc[j][qt][0] + nmx[qt], sc[j][qt][1]); e[qt][2 * j + 1] = round_pair<T>(sc[j][qt][2], sc[j][qt][3]); }
                else { e[qt][2 * j] = AttnExp<T>::pair(sc[j][qt][0], sc[j][qt][1], nmx[qt]); e[qt][2 * j + 1] = AttnExp<T>::pair(sc[j][qt][2], sc[j][qt][3], nmx[qt]); }
                sum[qt] = Pair<T>::sum2(e[qt][2 * j + 1], Pair<T>::sum2(e[qt][2 * j], sum[qt]));
            }
#pragma unroll
        for (int g = 0; g < 8; ++g) {            // one product, two transcendentals, four other VALU instructions, eight times
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x400, 2, 0); __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        read_v(sb, ks_, I0{}, I4{});                                                 // this step's V^T fragments, into the registers the products above have read
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) p_prev[qt] = v8{e[qt][0][0], e[qt][0][1], e[qt][1][0], e[qt][1][1], e[qt][2][0], e[qt][2][1], e[qt][3][0], e[qt][3][1]};
    };
    for (int c = 0; c < nch; ++c) {
        const int i = nch + c;
        if (AS_PARANOID) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
        if (i + AHEAD < nstage && !(FL & 16)) stage(i + AHEAD);
        __builtin_amdgcn_sched_barrier(0);
        if (active) {
            const unsigned sb = lds0 + (unsigned)((i % NSLOT) * SLOT);
            const int key0 = c * CK;
            if constexpr (PIPE) {
                if (c + 1 < nch) { p2_pipe(sb, key0, I0{}, FF{}, true); p2_pipe(sb, key0, I1{}, FF{}, false); }
                else { const bool two = key0 + 32 < N; p2_pipe(sb, key0, I0{}, TT{}, two); if (two) p2_pipe(sb, key0, I1{}, TT{}, false); }
                v_landed();               // before the barrier: the slot is re-staged right behind it, and the next chunk's first PV takes these fragments
            } else {
                if (c + 1 < nch) { p2_step(sb, key0, I0{}, FF{}); p2_step(sb, key0, I1{}, FF{}); }
                else { p2_step(sb, key0, I0{}, TT{}); if (key0 + 32 < N) p2_step(sb, key0, I1{}, TT{}); }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < nch && !(FL & 16)) {
            wait_vm_rt(ops_after(i));
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (!active) return;
    if constexpr (PIPE) pv(p_prev, I0{}, I4{});           // the last step's products
    // ---- O / sum, rounded once; lane (l15 = query, g4) holds O[query][dt * 16 + 4 g4 .. + 3].  v_permlane16_swap: the even lane row gives
    // its odd tile and takes the odd row's even tile -> 8 consecutive dims per lane, two 16-byte stores covering whole 64-byte lines
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    const int st_lane = (g4 & 1) * 16 + (g4 >> 1) * 8;          // elements
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const float inv = 1.0f / rows4_sum(sum[qt]);
        const int qrow = q0 + qt * 16 + l15;
        T *orow = out + ((size_t)b * N + min(qrow, N - 1)) * D + h * 64 + st_lane;
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            f32x4 oe = o[qt][2 * pr], oo = o[qt][2 * pr + 1];
            if constexpr (PREC) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { oe[r] = __builtin_fmaf(oc[qt][2 * pr][r], kHiLoInv, oe[r]); oo[r] = __builtin_fmaf(oc[qt][2 * pr + 1][r], kHiLoInv, oo[r]); }
            }
            const v2 elo = round_pair<T>(oe[0] * inv, oe[1] * inv), ehi = round_pair<T>(oe[2] * inv, oe[3] * inv);
            const v2 olo = round_pair<T>(oo[0] * inv, oo[1] * inv), ohi = round_pair<T>(oo[2] * inv, oo[3] * inv);
            const auto lo = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, elo), __builtin_bit_cast(unsigned, olo), false, false);
            const auto hi = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, ehi), __builtin_bit_cast(unsigned, ohi), false, false);
            const unsigned l0 = lo[0], l1 = lo[1], h0 = hi[0], h1 = hi[1];
            if (qrow < N) *(u32x4_t *)(orow + pr * 32) = u32x4_t{l0, h0, l1, h1};
        }
    }
}

// ------------------------------------------------------------------------------------------------
// attention_precise_kernel -- the F16 parity mode's f32-grade attention at 193..224 tokens (every headline model), PERSISTENT (r05).
// Same arithmetic, operation for operation, as the resident-score build of attention_stream_kernel it replaces (three MFMAs per score:
// hi.hi + (hi.lo + lo.hi) / 2048; RES score tiles of a query in registers between ONE pass over K and the pass over V; ggml's fp16 exp table;
// two MFMAs per P.V) -- the bits are the same.  What changed is the structure around it: that build was one workgroup per (image, head) item
// with its Q loads, its first two chunk fills and a drain exposed in front of every item (98 us per 128-image launch against ~60 us of traffic).
// Here one workgroup per CU walks the items and the operand stream never stops:
//   * 16 waves x 128 registers, wave w = the 16-query tile w of the item (13 compute for 197 tokens, the rest only move data);
//   * LDS = [Q: 4 chunks | ring: 6 slots] of 16 KiB = all 160 KiB.  A chunk is 64 rows of one operand as two plane images [hi 8 KiB | lo 8 KiB];
//     an item is 12 chunks: 4 of Q rows, 4 of K rows (swizzled row images: the fragment layout of the GEMMs), 4 of V rows (row-major, 32-byte
//     pieces XOR-ed with (row >> 1) & 3 for the transposed reads).  Q too comes by LDS-DMA -- no fragment registers are held for the next item;
//   * one STEP per K / V chunk (8 per item, one raw barrier each).  Every wave issues exactly ONE 16-byte-per-lane LDS-DMA instruction per
//     chunk (waves 0-7 the hi plane, 8-15 the lo plane), so all waves share one vmcnt schedule: the ring chunk that step t + 5 will read is
//     requested at the top of step t (its slot was released by the barrier behind step t - 1), and the next item's Q chunks in steps 1..4
//     (the Q region is free once every wave has taken its fragments in step 0).  The counted wait in front of a barrier names exactly the
//     operations younger than the chunk the next step reads (wait_cnt below) -- four to eight stay in flight across every barrier;
//   * the item's output (two 16-byte stores per lane) is issued at the end of step 7 and counted the same way.
// Keys past N: chunk rows beyond the item are the next item's rows or the zeros a buffer load returns out of range -- finite; their scores are
// masked to -inf (probability exactly 0).
// ------------------------------------------------------------------------------------------------
template <int RES>
__global__ __launch_bounds__(1024, 4) void attention_precise_kernel(const _Float16 *__restrict__ qkv, _Float16 *__restrict__ out, int N, int D, int H, int items, unsigned plane_bytes, long lo_off) {
    using namespace as;
    typedef _Float16 T;
    typedef Elem<T>::v8 v8;
    typedef Pair<T>::v2 v2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NKS = (RES + 1) / 2, NSL = 6, SLOT = 2 * KB, QB = 4 * SLOT;
    static_assert(RES == 13 || RES == 14, "193..224 tokens");
    asm volatile("v_mov_b32 v127, 0" ::: "v127");        // the whole register file of the SIMDs (AS_CLAIM, see attention_stream_kernel): 16 waves x 128
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row_bytes = 3 * D * 2;
    const bool active = wave * 16 < N;                  // this wave's query tile holds a real query
    const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)((__attribute__((address_space(3))) char *)smem);

    // ---- LDS-DMA: this wave moves 1 KiB of ONE plane of every chunk: pieces (wave & 7) * 64 + lane of the 512 a plane image has
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(qkv + ((wave >> 3) ? lo_off : 0)), 0, (int)plane_bytes, 0x00020000);
    // AP_DROP: the waves that move the lo plane (8..15) fetch nothing for an operand whose lo products are not formed -- a descriptor of zero records:
    // every lane is out of range, the buffer unit writes zeros and reads no memory, and the vmcnt schedule keeps its shape
    const bool lo_wave = (wave >> 3) != 0;
    const bool skip_k = (AP_DROP & 1) && lo_wave, skip_v = (AP_DROP & 2) && lo_wave, skip_q = (AP_DROP & 4) && lo_wave;
    int koff, voff;
    {
        const int piece = tid & 511;
        int rr, sl; swz_inv(piece, rr, sl);                                        // Q / K: swizzled row image, permutation on the source side
        koff = rr * row_bytes + sl * 16;
        const int vr = piece >> 3, vs = (piece & 7) ^ (((vr >> 1) & 3) << 1);      // V: row-major, 32-byte pieces XOR-ed with (row >> 1) & 3
        voff = vr * row_bytes + vs * 16;
    }
    const int dst_w = (wave >> 3) * KB + (wave & 7) * 1024;
    // (readfirstlane: the division runs on the VALU; its result must not sit in a vector register for the length of an item)
    auto item_base = [&](int item) -> unsigned { const int b = __builtin_amdgcn_readfirstlane(item / H), h = item - b * H; return (unsigned)(((size_t)b * N * 3 * D + h * 64) * 2); };
    int ring_in = 0, ring_out = 0;                       // slot the next ring chunk is written to / the current step reads
    // The fourth chunk of an operand covers rows 192..255 of the item; only rows < 16 RES (208 or 224) can hold a token.  Its DMA goes through a
    // descriptor that ENDS at row 16 RES of the item: the rows behind it are out of range -- the buffer unit returns zeros and fetches nothing
    // (a fifth of the launch's traffic; the zero rows are keys past N: masked, probability 0).
    const void *plane_ptr = (const void *)(qkv + ((wave >> 3) ? lo_off : 0));
    const unsigned tail_bytes = (unsigned)(RES * 16 * row_bytes);      // (a run-time value on purpose: hipcc 7.2 drops the host-side instantiation of a kernel whose lambda names the template parameter here)
    auto tail_end = [&](unsigned jb) -> int { const unsigned end = jb + tail_bytes; return (int)(end < plane_bytes ? end : plane_bytes); };     // (the descriptor itself is built at the use: a lambda RETURNING one makes hipcc 7.2 drop the kernel's host-side instantiation)
    auto stage_ring = [&](unsigned jb, int s) {          // chunk s of the item at byte offset jb: 0..3 = K, 4..7 = V
        const int so = __builtin_amdgcn_readfirstlane((int)(jb + (unsigned)((s & 3) * CK * row_bytes + (s < 4 ? D * 2 : 2 * D * 2))));
        const bool skip = AP_DROP && (s < 4 ? skip_k : skip_v);
        {
            __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc((void *)plane_ptr, 0, skip ? 0 : ((s & 3) == 3 ? tail_end(jb) : (int)plane_bytes), 0x00020000);
            if (AP_DROP || (s & 3) == 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rt, LPTR(smem + QB + ring_in * SLOT + dst_w), 16, s < 4 ? koff : voff, so, 0, AP_AUX);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LPTR(smem + QB + ring_in * SLOT + dst_w), 16, s < 4 ? koff : voff, so, 0, AP_AUX);
        }
        ring_in = ring_in == NSL - 1 ? 0 : ring_in + 1;
    };
    auto stage_q = [&](unsigned jb, int q) {
        const int so = __builtin_amdgcn_readfirstlane((int)(jb + (unsigned)(q * CK * row_bytes)));
        {
            __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc((void *)plane_ptr, 0, (AP_DROP && skip_q) ? 0 : (q == 3 ? tail_end(jb) : (int)plane_bytes), 0x00020000);
            if (AP_DROP || q == 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rt, LPTR(smem + q * SLOT + dst_w), 16, koff, so, 0, AP_AUX);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LPTR(smem + q * SLOT + dst_w), 16, koff, so, 0, AP_AUX);
        }
    };

    // ---- fragment addresses inside a chunk (attention_stream_kernel): K / Q tile t (16 rows) = parity (t & 1) base + (t >> 1) * 4096;
    // V step ks (32 keys) adds ks * 4096, its second 16 keys 2048.  They are loop-invariant per-lane values, and this kernel has no register to
    // keep them in across the phases that do not use them: each phase derives its own from an OPAQUE copy of the lane id (a handful of VALU
    // operations per item) -- left to itself hipcc hoists all of them in front of the item loop and spills score tiles instead.
    int krd[2][2], vrd[4];
    auto make_krd = [&]() {
        int l = lane; asm volatile("" : "+v"(l));
#pragma unroll
        for (int pz = 0; pz < 2; ++pz)
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) krd[pz][k2] = swz_byte(pz * 16 + (l & 15), k2 * 4 + (l >> 4));
    };
    auto make_vrd = [&]() {
        int l = lane; asm volatile("" : "+v"(l));
        const int r = 4 * (l >> 4) + ((l & 15) >> 2), x = (r >> 1) & 3;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) vrd[dt] = r * 128 + ((dt ^ x) << 5) + (l & 3) * 8;
    };
    const int q_chunk = (wave >> 2) * SLOT + ((wave & 3) >> 1) * 4096;      // this wave's 16 query rows inside the Q region: chunk wave >> 2, 16-row tile wave & 3 of it

    // K fragments of one 16-key tile (both planes); AP_K2: the step's second tile is requested into its own registers at the same time and
    // copied over when its turn comes.  The item loop must stay free of spills: a spill is a vector-memory operation inside a counted vmcnt schedule
    i4 kf[2], kl[2];
#if AP_K2
    i4 kf2[2], kl2[2];      // the second 16-key tile of a 32-key step is requested together with the first
#endif
    auto read_k = [&](unsigned sb, auto ks_, auto j_) {
        constexpr int ks = decltype(ks_)::value, j = decltype(j_)::value;
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) { ds_read_b128<ks * 4096>(kf[k2], sb + krd[j][k2]); ds_read_b128<ks * 4096 + KB>(kl[k2], sb + krd[j][k2]); }
    };
    v8 qh[2], ql[2];
    auto score = [&]() -> f32x4 {                        // S^T tile: lane (l15 = query, g4) gets keys 4 g4 .. + 3; hi.hi + (hi.lo + lo.hi) / 2048
        const v8 k0 = __builtin_bit_cast(v8, kf[0]), k1 = __builtin_bit_cast(v8, kf[1]);
        const v8 l0 = __builtin_bit_cast(v8, kl[0]), l1 = __builtin_bit_cast(v8, kl[1]);
        f32x4 a = {0.0f, 0.0f, 0.0f, 0.0f}, c = {0.0f, 0.0f, 0.0f, 0.0f};
        a = Elem<T>::mfma16(k0, qh[0], a);
        a = Elem<T>::mfma16(k1, qh[1], a);
        if constexpr ((AP_DROP & 4) == 0) {
            c = Elem<T>::mfma16(k0, ql[0], c);
            c = Elem<T>::mfma16(k1, ql[1], c);
        }
        if constexpr ((AP_DROP & 1) == 0) {
            c = Elem<T>::mfma16(l0, qh[0], c);
            c = Elem<T>::mfma16(l1, qh[1], c);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] = __builtin_fmaf(c[r], kHiLoInv, a[r]);
        return a;
    };
    s4 vf[4][2], vl[4][2];
    auto read_v = [&](unsigned sb, auto ks_, auto dt0_) {        // V^T fragments of 16-dim tiles dt0, dt0 + 1 of one 32-key step, both planes
        constexpr int ks = decltype(ks_)::value, DT0 = decltype(dt0_)::value;
        static_for<DT0, DT0 + 2>([&](auto dt_) {
            constexpr int dt = decltype(dt_)::value;
            ds_read_tr<ks * 4096>(vf[dt][0], sb + vrd[dt]); ds_read_tr<ks * 4096 + 2048>(vf[dt][1], sb + vrd[dt]);
            ds_read_tr<ks * 4096 + KB>(vl[dt][0], sb + vrd[dt]); ds_read_tr<ks * 4096 + 2048 + KB>(vl[dt][1], sb + vrd[dt]);
        });
    };
    auto pv = [&](f32x4 (&o)[4], f32x4 (&oc)[4], const v8 &p, auto dt0_) {
        constexpr int DT0 = decltype(dt0_)::value;
        static_for<DT0, DT0 + 2>([&](auto dt_) {
            constexpr int dt = decltype(dt_)::value;
            const s8 both = __builtin_shufflevector(vf[dt][0], vf[dt][1], 0, 1, 2, 3, 4, 5, 6, 7);
            o[dt] = Elem<T>::mfma16(__builtin_bit_cast(v8, both), p, o[dt]);
            const s8 bl = __builtin_shufflevector(vl[dt][0], vl[dt][1], 0, 1, 2, 3, 4, 5, 6, 7);
            if constexpr ((AP_DROP & 2) == 0) oc[dt] = Elem<T>::mfma16(__builtin_bit_cast(v8, bl), p, oc[dt]);
        });
    };
    typedef std::integral_constant<int, 0> I0; typedef std::integral_constant<int, 2> I2;
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

    // ---- prologue: the first item's Q chunks, then its first five ring chunks; Q and the first K chunk landed
    int item = blockIdx.x;
    if (item >= items) return;
    {
        const unsigned jb = item_base(item);
#pragma unroll
        for (int q = 0; q < 4; ++q) stage_q(jb, q);
#pragma unroll
        for (int s2 = 0; s2 < 5; ++s2) stage_ring(jb, s2);
    }
    wait_vm<4>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // The item loop exists twice, for the waves that own a query tile and for those that only move data: one run-time `if (active)` per step
    // would make every accumulator look live around the whole loop to the register allocator (42 spilled registers -- and a spill is a
    // vector-memory operation inside a counted vmcnt schedule).
    auto run = [&](auto active_) {
    constexpr bool ACTIVE = decltype(active_)::value;
    constexpr int nst = ACTIVE ? 2 : 0;                  // output stores of this wave per item (vector-memory operations like the DMA: counted)
    bool first = true;

    for (; item < items; item += gridDim.x) {
        const unsigned jb = item_base(item);
        const int nitem = item + (int)gridDim.x;
        const bool hn = nitem < items;
        const unsigned jbn = hn ? item_base(nitem) : 0u;
        const int hn1 = hn ? 1 : 0, pst = first ? 0 : nst;
        f32x4 sres[RES], o[4], oc[4];
        float mx = -INFINITY, nmx = 0.0f, sum = 0.0f;
        static_for<0, 8>([&](auto c_) {
            constexpr int c = decltype(c_)::value;
            // ---- top of the step: the slot the previous step read is free
            if constexpr (c <= 2) stage_ring(jb, 5 + c);
            else { if (hn) stage_ring(jbn, c - 3); }
            if constexpr (c >= 1 && c <= 4) { if (hn) stage_q(jbn, c - 1); }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (ACTIVE) {
                const unsigned sb = lds0 + (unsigned)(QB + ring_out * SLOT);
                if constexpr (c == 0) {       // this item's Q fragments (B operand of S^T = K . Q^T): lane (l15 = query, g4) holds dims k2 * 32 + g4 * 8 .. + 7
                    make_krd();
                    const unsigned qaddr0 = lds0 + (unsigned)(q_chunk + ((wave & 1) ? krd[1][0] : krd[0][0])), qaddr1 = lds0 + (unsigned)(q_chunk + ((wave & 1) ? krd[1][1] : krd[0][1]));
                    i4 t0, t1, t2, t3;
                    ds_read_b128<0>(t0, qaddr0); ds_read_b128<0>(t1, qaddr1);
                    ds_read_b128<KB>(t2, qaddr0); ds_read_b128<KB>(t3, qaddr1);
                    wait_lgkm4<0>(t0, t1, t2, t3);
                    qh[0] = __builtin_bit_cast(v8, t0); qh[1] = __builtin_bit_cast(v8, t1); ql[0] = __builtin_bit_cast(v8, t2); ql[1] = __builtin_bit_cast(v8, t3);
                }
                if constexpr (c < 4) {        // K chunk c: score tiles 4 c .. 4 c + 3
                    static_for<0, 2>([&](auto ks_) {
                        constexpr int ks = decltype(ks_)::value;
                        if constexpr (4 * c + 2 * ks < RES) {
                            static_for<0, 2>([&](auto j_) {
                                constexpr int j = decltype(j_)::value, t = 4 * c + 2 * ks + j;
                                if constexpr (t < RES) {
#if AP_K2
                                    if constexpr (j == 0) {
                                        read_k(sb, ks_, j_);
                                        if constexpr (t + 1 < RES) {
#pragma unroll
                                            for (int k2 = 0; k2 < 2; ++k2) { ds_read_b128<ks * 4096>(kf2[k2], sb + krd[1][k2]); ds_read_b128<ks * 4096 + KB>(kl2[k2], sb + krd[1][k2]); }
                                            wait_lgkm4<4>(kf[0], kf[1], kl[0], kl[1]);
                                        } else wait_lgkm4<0>(kf[0], kf[1], kl[0], kl[1]);
                                    } else {
                                        wait_lgkm4<0>(kf2[0], kf2[1], kl2[0], kl2[1]);
                                        kf[0] = kf2[0]; kf[1] = kf2[1]; kl[0] = kl2[0]; kl[1] = kl2[1];
                                    }
#else
                                    read_k(sb, ks_, j_);
                                    wait_lgkm4<0>(kf[0], kf[1], kl[0], kl[1]);
#endif
                                    f32x4 sc = score();
                                    if constexpr (t >= 12) {            // only the tiles past key 191 can hold padded keys (N > 192)
                                        int l = lane; asm volatile("" : "+v"(l));
#pragma unroll
                                        for (int r = 0; r < 4; ++r) if (t * 16 + 4 * (l >> 4) + r >= N) sc[r] = -INFINITY;
                                    }
                                    asm volatile("" : "+v"(sc));      // materialise NOW: with its first use steps away LLVM sinks the hi/lo combine and keeps BOTH accumulators of every tile (8 registers a tile: spills)
                                    sres[t] = sc;
                                    mx = fmaxf(fmaxf(mx, sc[0]), sc[1]); mx = fmaxf(fmaxf(mx, sc[2]), sc[3]);
                                }
                            });
                        }
                    });
                    if constexpr (c == 3) {
                        nmx = -AttnExp<T>::kScale * rows4_max(mx);
#pragma unroll
                        for (int dt = 0; dt < 4; ++dt) { o[dt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; oc[dt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
                    }
                } else {                      // V chunk c - 4: 32-key steps 2 (c - 4), 2 (c - 4) + 1
                    if constexpr (c == 4) make_vrd();
                    static_for<0, 2>([&](auto ks_) {
                        constexpr int ks = decltype(ks_)::value, t2 = 2 * (c - 4) + ks;
                        if constexpr (t2 < NKS) {
                            read_v(sb, ks_, I0{});                                     // dims 0..31, both planes: in flight under the exp arithmetic
                            const v2 e0 = AttnExp<T>::pair(sres[2 * t2][0], sres[2 * t2][1], nmx), e1 = AttnExp<T>::pair(sres[2 * t2][2], sres[2 * t2][3], nmx);
                            v2 e2 = __builtin_bit_cast(v2, 0u), e3 = __builtin_bit_cast(v2, 0u);            // an odd last tile: probabilities 0 (its V rows are finite)
                            sum = Pair<T>::sum2(e1, Pair<T>::sum2(e0, sum));
                            if constexpr (2 * t2 + 1 < RES) {
                                e2 = AttnExp<T>::pair(sres[2 * t2 + 1][0], sres[2 * t2 + 1][1], nmx); e3 = AttnExp<T>::pair(sres[2 * t2 + 1][2], sres[2 * t2 + 1][3], nmx);
                                sum = Pair<T>::sum2(e3, Pair<T>::sum2(e2, sum));
                            }
                            const v8 p = v8{e0[0], e0[1], e1[0], e1[1], e2[0], e2[1], e3[0], e3[1]};
                            wait_lgkm8<0>(vf[0][0], vf[0][1], vf[1][0], vf[1][1], vl[0][0], vl[0][1], vl[1][0], vl[1][1]);
                            pv(o, oc, p, I0{});
                            read_v(sb, ks_, I2{});
                            wait_lgkm8<0>(vf[2][0], vf[2][1], vf[3][0], vf[3][1], vl[2][0], vl[2][1], vl[3][0], vl[3][1]);
                            pv(o, oc, p, I2{});
                        }
                    });
                    if constexpr (c == 7) {   // O / sum, rounded once (attention_stream_kernel's tail, the same operations)
                        const int b = __builtin_amdgcn_readfirstlane(item / H), h = item - b * H;
                        const float inv = 1.0f / rows4_sum(sum);
                        int l = lane; asm volatile("" : "+v"(l));
                        const int qrow = wave * 16 + (l & 15);
                        const int st_lane = ((l >> 4) & 1) * 16 + (l >> 5) * 8;          // elements: after the row swap a lane holds dims 8 (g4 >> 1) .. + 7 of head-dim tile 2 pr + (g4 & 1)
                        T *orow = out + ((size_t)b * N + min(qrow, N - 1)) * D + h * 64 + st_lane;
#pragma unroll
                        for (int pr = 0; pr < 2; ++pr) {
                            f32x4 oe = o[2 * pr], oo = o[2 * pr + 1];
#pragma unroll
                            for (int r = 0; r < 4; ++r) { oe[r] = __builtin_fmaf(oc[2 * pr][r], kHiLoInv, oe[r]); oo[r] = __builtin_fmaf(oc[2 * pr + 1][r], kHiLoInv, oo[r]); }
                            const v2 elo = round_pair<T>(oe[0] * inv, oe[1] * inv), ehi = round_pair<T>(oe[2] * inv, oe[3] * inv);
                            const v2 olo = round_pair<T>(oo[0] * inv, oo[1] * inv), ohi = round_pair<T>(oo[2] * inv, oo[3] * inv);
                            const auto lo = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, elo), __builtin_bit_cast(unsigned, olo), false, false);
                            const auto hi = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, ehi), __builtin_bit_cast(unsigned, ohi), false, false);
                            const unsigned l0 = lo[0], l1 = lo[1], h0 = hi[0], h1 = hi[1];
                            // lanes whose query lies past N are masked by exec; the instruction still issues once per wave (lane 0 of an active wave holds a
                            // real query), which is what the vmcnt schedule counts: two stores per active wave and item
                            if (qrow < N) *(u32x4_t *)(orow + pr * 32) = u32x4_t{l0, h0, l1, h1};
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- the chunk the next step reads has landed; what may stay in flight = the operations issued after it (see the header):
            // ring requests of the last four steps, the next item's Q requests among them, this wave's output stores
            if constexpr (c == 7) { if (hn) wait_vm_rt(3 + nst); }         // + every Q chunk of the next item (requested in steps 1..4)
            else {
                constexpr int ring_a = c <= 2 ? 4 : (c == 3 ? 3 : (c == 4 ? 2 : (c == 5 ? 1 : 0))), ring_h = c <= 2 ? 0 : (c == 3 ? 1 : (c == 4 ? 2 : (c == 5 ? 3 : 4)));
                constexpr int q_h = c == 0 ? 0 : (c <= 4 ? c : (c == 5 ? 4 : 3));
                wait_vm_rt(ring_a + (ring_h + q_h) * hn1 + (c == 0 && !first ? 1 : 0) + (c <= 3 ? pst : 0));
            }
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            ring_out = ring_out == NSL - 1 ? 0 : ring_out + 1;
        });
        first = false;
    }
    };
    if (active) run(std::true_type{}); else run(std::false_type{});
}

static hipError_t check_register_claim(const void *kernel, int threads, int lds, int want_regs);
template <typename T, int QT, bool PREC>
static hipError_t launch_stream_inst(const void *qkv, void *out, int n_img, int N, int D, int H, long lo_off, hipStream_t stream) {
    constexpr int W = as::nwaves<PREC>(), NSLOT = as::nslot<PREC>(), lds = NSLOT * as::slot_bytes<PREC>();
    if (n_img == 0) {      // device bring-up
        hipError_t e = hipFuncSetAttribute((const void *)attention_stream_kernel<T, QT, PREC, W, NSLOT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
#if AS_CLAIM
        if (e == hipSuccess) e = check_register_claim((const void *)attention_stream_kernel<T, QT, PREC, W, NSLOT>, W * 64, lds, W == 16 ? 128 : 256);
#endif
        return e;
    }
    const int tasks = (N + 16 * QT - 1) / (16 * QT), qblocks = (tasks + W - 1) / W, items = n_img * H;
    const unsigned grid = (unsigned)(((items + 7) / 8) * 8 * qblocks);
    hipLaunchKernelGGL((attention_stream_kernel<T, QT, PREC, W, NSLOT>), dim3(grid), dim3(W * 64), lds, stream, (const T *)qkv, (T *)out, N, D, H, items, qblocks, n_img, lo_off);
    return hipGetLastError();
}

// Bring-up check of the register claim (r04 advisor): these kernels must own the whole register file of their SIMDs (a foreign wave beside
// their MFMA stream computed wrong DPP sums: profiles/r04/coresidency_layernorm.txt).  The claim is an inline v_mov to the top register;
// a compiler change that allocates differently, or another register-file size, would silently bring the hazard back -- so the allocation the
// code object reports (granule 8) and the occupancy the runtime computes are checked when the device is brought up, and a mismatch fails it.
static hipError_t check_register_claim(const void *kernel, int threads, int lds, int want_regs) {
    hipFuncAttributes fa;
    hipError_t e = hipFuncGetAttributes(&fa, kernel);
    if (e != hipSuccess) return e;
    int blocks = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, kernel, threads, (size_t)lds);
    if (e != hipSuccess) return e;
    if ((fa.numRegs + 7) / 8 * 8 != want_regs || blocks != 1) return hipErrorLaunchOutOfResources;
    return hipSuccess;
}

template <int RES>
static hipError_t launch_precise_inst(const void *qkv, void *out, int n_img, int N, int D, int H, long lo_off, hipStream_t stream) {
    constexpr int lds = 10 * 2 * as::KB;          // 4 Q chunks + 6 ring slots of 16 KiB
    if (n_img == 0) {      // device bring-up
        hipError_t e = hipFuncSetAttribute((const void *)attention_precise_kernel<RES>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e == hipSuccess) e = check_register_claim((const void *)attention_precise_kernel<RES>, 1024, lds, 128);
        return e;
    }
    const size_t plane = (size_t)n_img * N * 3 * D * 2;
    if (plane > 0xf0000000u || (RES * 16 < N) || N <= (RES - 1) * 16) return hipErrorInvalidValue;
    const Tuning *t = tuning_for_device(-1);
    const int items = n_img * H, cus = t ? t->n_cu : 256;
    const int grid = items < cus ? items : cus;
    hipLaunchKernelGGL((attention_precise_kernel<RES>), dim3(grid), dim3(1024), lds, stream, (const _Float16 *)qkv, (_Float16 *)out, N, D, H, items, (unsigned)plane, lo_off);
    return hipGetLastError();
}

bool attention_stream_supports(int n_img, int N, int D, int H) { return N > 0 && H > 0 && D == H * 64 && n_img >= 0; }

// precise = the F16 parity mode's f32-grade products: qkv holds the hi plane, the lo plane lies lo_off ELEMENTS behind it (EPI_BIAS_HILO)
hipError_t launch_attention_stream(int dtype, bool precise, const void *qkv, void *out, int n_img, int N, int D, int H, long lo_off, hipStream_t stream) {
    if (n_img != 0 && !attention_stream_supports(n_img, N, D, H)) return hipErrorInvalidValue;
    if (precise) {
        if (dtype != DT_F16) return hipErrorInvalidValue;
        if (n_img == 0) {       // bring-up: every precise build
            hipError_t e = launch_stream_inst<_Float16, 1, true>(qkv, out, 0, N, D, H, lo_off, stream);
            if (e == hipSuccess) e = launch_precise_inst<13>(qkv, out, 0, N, D, H, lo_off, stream);
            if (e == hipSuccess) e = launch_precise_inst<14>(qkv, out, 0, N, D, H, lo_off, stream);
            return e;
        }
        // 193..224 tokens (ViT-*/16 at 224^2): the persistent kernel -- the score tiles stay in registers, K is streamed once, one workgroup per
        // CU walks the items.  One build per token count, at every batch size: an image's result does not depend on the batch it arrives in.
        if (N > 192 && N <= 208) return launch_precise_inst<13>(qkv, out, n_img, N, D, H, lo_off, stream);
        if (N > 208 && N <= 224) return launch_precise_inst<14>(qkv, out, n_img, N, D, H, lo_off, stream);
        return launch_stream_inst<_Float16, 1, true>(qkv, out, n_img, N, D, H, lo_off, stream);
    }
    return dtype == DT_F16 ? launch_stream_inst<_Float16, 2, false>(qkv, out, n_img, N, D, H, 0, stream) : launch_stream_inst<__bf16, 2, false>(qkv, out, n_img, N, D, H, 0, stream);
}

// ------------------------------------------------------------------------------------------------
// f32 [rows][cols] -> the two 16-bit planes the precise kernel reads (the same split the QKV GEMM's EPI_BIAS_HILO epilogue emits):
// hi[i] = round(x), lo[i] = round((x - hi) * 2048).  Used by the single-kernel entry point of the parity tests (vitx_op_attention_f32).
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void split_hilo_kernel(const float *__restrict__ x, T *__restrict__ hi, T *__restrict__ lo, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float v = x[i];
        const T h = (T)v;
        hi[i] = h; lo[i] = (T)((v - (float)h) * kHiLoScale);
    }
}
hipError_t launch_split_hilo(int dtype, const float *x, void *hi, void *lo, size_t n, hipStream_t stream) {
    const unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 4096);
    if (dtype == DT_F16) hipLaunchKernelGGL(split_hilo_kernel<_Float16>, dim3(grid), dim3(256), 0, stream, x, (_Float16 *)hi, (_Float16 *)lo, n);
    else hipLaunchKernelGGL(split_hilo_kernel<__bf16>, dim3(grid), dim3(256), 0, stream, x, (__bf16 *)hi, (__bf16 *)lo, n);
    return hipGetLastError();
}

}  // namespace vitx

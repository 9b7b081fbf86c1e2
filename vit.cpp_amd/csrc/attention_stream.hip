// attention_stream.hip -- streaming two-pass attention for gfx950 (vit.cpp:826-866), any token count, head dim 64.
//
// Two builds of one kernel:
//   * fast  (QT = 2, PREC = false; f16 or bf16 operands): the long-sequence kernel (577 tokens of ViT-L/16-384).  r03's pipelined kernel
//     (attention_flow_kernel, kernels.hip) ran 0.20 of the MFMA peak: 32x32x16 products, a vmcnt(0) + __syncthreads per 64-key chunk and
//     compiler-scheduled LDS reads that drain the in-flight LDS-DMA.  Here: v_mfma_f32_16x16x32 (the GEMMs' instruction), a wave owns
//     TWO 16-query tiles so every K / V^T fragment read from LDS feeds two products (one fragment per product is exactly the LDS peak:
//     1 KiB per 16-cycle MFMA per SIMD = 256 B/clk/CU), a THREE-slot ring of 64-key chunks filled by LDS-DMA two chunks ahead with
//     counted vmcnt and one raw barrier per chunk, every LDS read inline asm with counted lgkmcnt, row reductions by permlane swaps,
//     16-byte output stores; <= 128 VGPRs and 48 KiB of LDS, so two workgroups (16 waves) share a CU.
//   * precise (QT = 1, PREC = true; f16 only): the F16 PARITY MODE at every token count.  The reference multiplies f32 q, k, v
//     (ggml_mul_mat on f32 views, vit.cpp:848,858); r03 rounded them to fp16 for the MFMAs -- the one known semantic deviation of that
//     mode.  Here q, k, v arrive as TWO fp16 planes from the QKV GEMM (EPI_BIAS_HILO: hi = round(x), lo = round((x - hi) * 2048)) and
//     every product is three MFMAs, hi.hi + (hi.lo + lo.hi) / 2048 (the dropped lo.lo term is 2^-22 relative): f32-grade scores; the
//     probabilities are the fp16 exp-table values (exact in fp16, as ggml's LUT emits them), so P.V needs only V split: two MFMAs.
// Both: pass 1 streams K for the row maxima of the raw scores, pass 2 streams K and V: e = AttnExp<T>(s, max) (F16: ggml_soft_max's
// table semantics), row sum of the rounded numerators, O^T = V^T . P^T, O / sum rounded once to the operand type.
// Keys past N inside the last chunk read the next image's rows (finite; masked to -inf, probability exactly 0) or the zeros a buffer
// load returns out of range.
#include <type_traits>

#include "device_common.h"
#include "kernels.h"

namespace vitx {

namespace as {
constexpr int CK = 64;               // keys per chunk
constexpr int KB = CK * 128;         // bytes of one 64-row plane image (K or V rows of 64 dims x 2 B)
constexpr int NSLOT = 3;
template <bool PREC> constexpr int slot_bytes() { return (PREC ? 4 : 2) * KB; }      // [K hi | K lo | V hi | V lo] or [K | V]
template <int I, int N, typename F> __device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_vm_rt(int n) {       // n = DMA instructions of the youngest stage: 0, 1, 2 or 4
    if (n == 0) wait_vm<0>(); else if (n == 1) wait_vm<1>(); else if (n == 2) wait_vm<2>(); else wait_vm<4>();
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

template <typename T, int QT, bool PREC>
__global__ __launch_bounds__(512, PREC ? 2 : 4) void attention_stream_kernel(const T *__restrict__ qkv, T *__restrict__ out, int N, int D, int H, int items, int qblocks, int n_img, long lo_off) {
    using namespace as;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PL = PREC ? 2 : 1, SLOT = slot_bytes<PREC>();
    typedef typename Elem<T>::v8 v8;
    typedef typename Pair<T>::v2 v2;
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
    int koff, voff;
    {
        int rr, sl; swz_inv(tid, rr, sl);                                          // K: swizzled row image (swz_byte), permutation on the source side
        koff = rr * row_bytes + D * 2 + sl * 16;
        const int vr = tid >> 3, vs = (tid & 7) ^ (((vr >> 1) & 3) << 1);          // V: row-major, 32-byte chunks XOR-ed with (row >> 1) & 3
        voff = vr * row_bytes + 2 * D * 2 + vs * 16;
    }
    auto stage = [&](int i) {            // stage i: pass-1 chunk i (K only) for i < nch, pass-2 chunk i - nch (K and V) after that
        const bool with_v = i >= nch;
        const int c = with_v ? i - nch : i;
        char *dst = smem + (i % NSLOT) * SLOT + wave * 1024;
        const int so = c * CK * row_bytes;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LPTR(dst), 16, koff, so, 0, 0);
        if (PREC) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_lo, LPTR(dst + KB), 16, koff, so, 0, 0);
        if (with_v) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LPTR(dst + PL * KB), 16, voff, so, 0, 0);
            if (PREC) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_lo, LPTR(dst + PL * KB + KB), 16, voff, so, 0, 0);
        }
    };
    auto stage_ops = [&](int i) { return i >= nstage ? 0 : (i >= nch ? 2 * PL : PL); };

    // ---- Q fragments (B operand of S^T = K . Q^T): lane (l15 = query of the tile, g4) holds dims k2 * 32 + g4 * 8 .. + 7
    v8 qh[QT][2], ql[QT][2];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int qrow = min(q0 + qt * 16 + l15, N - 1);
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            qh[qt][k2] = *(const v8 *)(base + (size_t)qrow * 3 * D + k2 * 32 + g4 * 8);
            ql[qt][k2] = PREC ? *(const v8 *)(base + lo_off + (size_t)qrow * 3 * D + k2 * 32 + g4 * 8) : qh[qt][k2];
        }
    }
    stage(0);
    if (nstage > 1) stage(1);
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
        for (int dt = 0; dt < 4; ++dt) vrd[dt] = PL * KB + r * 128 + ((dt ^ x) << 5) + (l15 & 3) * 8;
    }

    // K fragments of one 32-key step: [16-key tile j][k2] (+ the lo plane)
    i4 kf[2][2], kl[2][2];
    auto read_k = [&](unsigned sb, auto ks_) {
        constexpr int ks = decltype(ks_)::value;
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            ds_read_b128<ks * 4096>(kf[0][k2], sb + krd[0][k2]); ds_read_b128<ks * 4096>(kf[1][k2], sb + krd[1][k2]);
            if constexpr (PREC) { ds_read_b128<ks * 4096 + KB>(kl[0][k2], sb + krd[0][k2]); ds_read_b128<ks * 4096 + KB>(kl[1][k2], sb + krd[1][k2]); }
        }
    };
    auto wait_k = [&](auto cnt_) {       // the K reads have landed; cnt = LDS reads issued after them
        constexpr int cnt = decltype(cnt_)::value;
        if constexpr (PREC) wait_lgkm8<cnt>(kf[0][0], kf[0][1], kf[1][0], kf[1][1], kl[0][0], kl[0][1], kl[1][0], kl[1][1]);
        else wait_lgkm4<cnt>(kf[0][0], kf[0][1], kf[1][0], kf[1][1]);
    };
    // S^T tile: rows = 16 keys, cols = the 16 queries of tile qt; lane (l15, g4) gets keys 4 g4 .. + 3 of query l15
    auto score = [&](int j, int qt) -> f32x4 {
        const v8 k0 = __builtin_bit_cast(v8, kf[j][0]), k1 = __builtin_bit_cast(v8, kf[j][1]);
        f32x4 a = {0.0f, 0.0f, 0.0f, 0.0f};
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
    typedef std::true_type TT; typedef std::false_type FF;
    for (int c = 0; c < nch; ++c) {
        const int i = c;
        if (i + 2 < nstage) stage(i + 2);
        __builtin_amdgcn_sched_barrier(0);
        if (active) {
            const unsigned sb = lds0 + (unsigned)((i % NSLOT) * SLOT);
            const int key0 = c * CK;
            if (c + 1 < nch) { p1_step(sb, key0, I0{}, FF{}); p1_step(sb, key0, I1{}, FF{}); }
            else { p1_step(sb, key0, I0{}, TT{}); if (key0 + 32 < N) p1_step(sb, key0, I1{}, TT{}); }
        }
        __builtin_amdgcn_sched_barrier(0);
        wait_vm_rt(stage_ops(i + 2));                   // stage i + 1 has landed; the pieces of stage i + 2 may stay in flight
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
        static_for<DT0, DT0 + NDT>([&](auto dt_) {
            constexpr int dt = decltype(dt_)::value;
            ds_read_tr<ks * 4096>(vf[dt][0], sb + vrd[dt]); ds_read_tr<ks * 4096 + 2048>(vf[dt][1], sb + vrd[dt]);
            if constexpr (PREC) { ds_read_tr<ks * 4096 + KB>(vl[dt][0], sb + vrd[dt]); ds_read_tr<ks * 4096 + 2048 + KB>(vl[dt][1], sb + vrd[dt]); }
        });
    };
    auto pv = [&](const v8 (&p)[QT], auto dt0_, auto ndt_) {
        constexpr int DT0 = decltype(dt0_)::value, NDT = decltype(ndt_)::value;
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
    typedef std::integral_constant<int, 2> I2; typedef std::integral_constant<int, 4> I4; typedef std::integral_constant<int, 8> I8;
    auto p2_step = [&](unsigned sb, int key0, auto ks_, auto masked_) {
        constexpr bool MASK = decltype(masked_)::value;
        constexpr int ks = decltype(ks_)::value;
        read_k(sb, ks_);
        if constexpr (!PREC) { read_v(sb, ks_, I0{}, I4{}); wait_k(I8{}); }          // 4 + 8 LDS reads in flight (the counter holds 15)
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
                e[qt][2 * j] = AttnExp<T>::pair(sc[0], sc[1], nmx[qt]); e[qt][2 * j + 1] = AttnExp<T>::pair(sc[2], sc[3], nmx[qt]);
                sum[qt] = Pair<T>::sum2(e[qt][2 * j + 1], Pair<T>::sum2(e[qt][2 * j], sum[qt]));
            }
            if (PREC && j == 0) read_v(sb, ks_, I0{}, I2{});                           // 8 reads: dims 0..31, both planes (the K fragments of tile 1 are in registers already)
        }
        v8 p[QT];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) p[qt] = v8{e[qt][0][0], e[qt][0][1], e[qt][1][0], e[qt][1][1], e[qt][2][0], e[qt][2][1], e[qt][3][0], e[qt][3][1]};
        if constexpr (!PREC) {
            wait_lgkm8<0>(vf[0][0], vf[0][1], vf[1][0], vf[1][1], vf[2][0], vf[2][1], vf[3][0], vf[3][1]);
            pv(p, I0{}, I4{});
        } else {
            wait_lgkm8<0>(vf[0][0], vf[0][1], vf[1][0], vf[1][1], vl[0][0], vl[0][1], vl[1][0], vl[1][1]);
            pv(p, I0{}, I2{});
            read_v(sb, ks_, I2{}, I2{});
            wait_lgkm8<0>(vf[2][0], vf[2][1], vf[3][0], vf[3][1], vl[2][0], vl[2][1], vl[3][0], vl[3][1]);
            pv(p, I2{}, I2{});
        }
    };
    for (int c = 0; c < nch; ++c) {
        const int i = nch + c;
        if (i + 2 < nstage) stage(i + 2);
        __builtin_amdgcn_sched_barrier(0);
        if (active) {
            const unsigned sb = lds0 + (unsigned)((i % NSLOT) * SLOT);
            const int key0 = c * CK;
            if (c + 1 < nch) { p2_step(sb, key0, I0{}, FF{}); p2_step(sb, key0, I1{}, FF{}); }
            else { p2_step(sb, key0, I0{}, TT{}); if (key0 + 32 < N) p2_step(sb, key0, I1{}, TT{}); }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < nch) {
            wait_vm_rt(stage_ops(i + 2));
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (!active) return;
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

template <typename T, int QT, bool PREC>
static hipError_t launch_stream_inst(const void *qkv, void *out, int n_img, int N, int D, int H, long lo_off, hipStream_t stream) {
    constexpr int lds = as::NSLOT * as::slot_bytes<PREC>();
    if (n_img == 0) return hipFuncSetAttribute((const void *)attention_stream_kernel<T, QT, PREC>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);      // device bring-up
    const int tasks = (N + 16 * QT - 1) / (16 * QT), qblocks = (tasks + 7) / 8, items = n_img * H;
    const unsigned grid = (unsigned)(((items + 7) / 8) * 8 * qblocks);
    hipLaunchKernelGGL((attention_stream_kernel<T, QT, PREC>), dim3(grid), dim3(512), lds, stream, (const T *)qkv, (T *)out, N, D, H, items, qblocks, n_img, lo_off);
    return hipGetLastError();
}

bool attention_stream_supports(int n_img, int N, int D, int H) { return N > 0 && H > 0 && D == H * 64 && n_img >= 0; }

// precise = the F16 parity mode's f32-grade products: qkv holds the hi plane, the lo plane lies lo_off ELEMENTS behind it (EPI_BIAS_HILO)
hipError_t launch_attention_stream(int dtype, bool precise, const void *qkv, void *out, int n_img, int N, int D, int H, long lo_off, hipStream_t stream) {
    if (n_img != 0 && !attention_stream_supports(n_img, N, D, H)) return hipErrorInvalidValue;
    if (precise) return dtype == DT_F16 ? launch_stream_inst<_Float16, 1, true>(qkv, out, n_img, N, D, H, lo_off, stream) : hipErrorInvalidValue;
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

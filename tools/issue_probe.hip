// How MFMA and VALU / transcendental instructions share a gfx950 SIMD: cycles per loop iteration (s_memtime) of fixed
// instruction mixes, for one wave per SIMD and for two co-resident waves per SIMD running DIFFERENT mixes.
//   hipcc --offload-arch=gfx950 -O2 -o issue_probe.bin issue_probe.hip ; ./issue_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CLOB "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63", \
             "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31", \
             "a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63"
#define M0 "v_mfma_f32_32x32x16_bf16 a[0:15], v[32:35], v[36:39], a[0:15]\n"
#define M1 "v_mfma_f32_32x32x16_bf16 a[16:31], v[32:35], v[36:39], a[16:31]\n"
#define M2 "v_mfma_f32_32x32x16_bf16 a[32:47], v[32:35], v[36:39], a[32:47]\n"
#define M3 "v_mfma_f32_32x32x16_bf16 a[48:63], v[32:35], v[36:39], a[48:63]\n"
#define F8 "v_fma_f32 v40, v40, v40, v40\nv_fma_f32 v41, v41, v41, v41\nv_fma_f32 v42, v42, v42, v42\nv_fma_f32 v43, v43, v43, v43\nv_fma_f32 v44, v44, v44, v44\nv_fma_f32 v45, v45, v45, v45\nv_fma_f32 v46, v46, v46, v46\nv_fma_f32 v47, v47, v47, v47\n"
#define E4 "v_exp_f32 v48, v48\nv_exp_f32 v49, v49\nv_exp_f32 v50, v50\nv_exp_f32 v51, v51\n"
#define C4 "v_cvt_pk_bf16_f32 v52, v52, v53\nv_cvt_pk_bf16_f32 v54, v54, v55\nv_cvt_pk_bf16_f32 v56, v56, v57\nv_cvt_pk_bf16_f32 v58, v58, v59\n"
#define P4 "v_pk_fma_f32 v[52:53], v[52:53], v[54:55], v[56:57]\nv_pk_fma_f32 v[54:55], v[54:55], v[56:57], v[58:59]\nv_pk_fma_f32 v[56:57], v[56:57], v[58:59], v[60:61]\nv_pk_fma_f32 v[58:59], v[58:59], v[60:61], v[62:63]\n"
#define D4 "v_dot2c_f32_bf16 v60, v52, v53\nv_dot2c_f32_bf16 v61, v54, v55\nv_dot2c_f32_bf16 v62, v56, v57\nv_dot2c_f32_bf16 v63, v58, v59\n"

#define K8 "v_fmamk_f32 v40, v40, 0x3e38aa3b, v48\nv_fmamk_f32 v41, v41, 0x3e38aa3b, v48\nv_fmamk_f32 v42, v42, 0x3e38aa3b, v48\nv_fmamk_f32 v43, v43, 0x3e38aa3b, v48\nv_fmamk_f32 v44, v44, 0x3e38aa3b, v48\nv_fmamk_f32 v45, v45, 0x3e38aa3b, v48\nv_fmamk_f32 v46, v46, 0x3e38aa3b, v48\nv_fmamk_f32 v47, v47, 0x3e38aa3b, v48\n"
#define A4 "v_pk_add_f32 v[40:41], v[40:41], v[48:49]\nv_pk_add_f32 v[42:43], v[42:43], v[48:49]\nv_pk_add_f32 v[44:45], v[44:45], v[48:49]\nv_pk_add_f32 v[46:47], v[46:47], v[48:49]\n"
#define X4 "v_max3_f32 v40, v40, v50, v51\nv_max3_f32 v41, v41, v52, v53\nv_max3_f32 v42, v42, v54, v55\nv_max3_f32 v43, v43, v56, v57\n"
#define E8 "v_exp_f32 v40, v40\nv_exp_f32 v41, v41\nv_exp_f32 v42, v42\nv_exp_f32 v43, v43\nv_exp_f32 v44, v44\nv_exp_f32 v45, v45\nv_exp_f32 v46, v46\nv_exp_f32 v47, v47\n"
#define H8 "v_exp_f16 v40, v40\nv_exp_f16 v41, v41\nv_exp_f16 v42, v42\nv_exp_f16 v43, v43\nv_exp_f16 v44, v44\nv_exp_f16 v45, v45\nv_exp_f16 v46, v46\nv_exp_f16 v47, v47\n"
#define C4B "v_cvt_pk_bf16_f32 v52, v40, v41\nv_cvt_pk_bf16_f32 v53, v42, v43\nv_cvt_pk_bf16_f32 v54, v44, v45\nv_cvt_pk_bf16_f32 v55, v46, v47\n"
#define D2 "v_dot2c_f32_bf16 v60, v52, v53\nv_dot2c_f32_bf16 v61, v54, v55\n"
template <int MODE> __device__ __forceinline__ void body() {
    if constexpr (MODE == 12) asm volatile(K8 K8 K8 K8 ::: CLOB);                                  // 32 v_fmamk
    if constexpr (MODE == 13) asm volatile(A4 A4 A4 A4 ::: CLOB);                                  // 16 v_pk_add_f32 (32 elements)
    if constexpr (MODE == 14) asm volatile(X4 X4 X4 X4 ::: CLOB);                                  // 16 v_max3
    if constexpr (MODE == 15) asm volatile(E8 E8 ::: CLOB);                                        // 16 v_exp_f32, 8 registers (dependent at distance 8)
    if constexpr (MODE == 16) asm volatile(H8 H8 ::: CLOB);                                        // 16 v_exp_f16
    if constexpr (MODE == 17) asm volatile(K8 E8 C4B D2 K8 E8 C4B D2 ::: CLOB);                    // softmax body for 16 scores: fmamk, exp, cvt, dot2c
    if constexpr (MODE == 18) asm volatile(A4 E8 C4B D2 A4 E8 C4B D2 ::: CLOB);                    // the same with packed adds
    if constexpr (MODE == 19) asm volatile(C4B C4B C4B C4B ::: CLOB);                              // 16 cvt_pk
    if constexpr (MODE == 20) asm volatile(D2 D2 D2 D2 D2 D2 D2 D2 ::: CLOB);                      // 16 dot2c (2 chains)

    if constexpr (MODE == 0) asm volatile(M0 M1 M2 M3 ::: CLOB);                                   // 4 independent MFMAs
    if constexpr (MODE == 1) asm volatile(M0 F8 M1 F8 M2 F8 M3 F8 ::: CLOB);                       // interleaved with 32 v_fma
    if constexpr (MODE == 2) asm volatile(F8 F8 F8 F8 ::: CLOB);                                   // 32 v_fma alone
    if constexpr (MODE == 3) asm volatile(M0 E4 M1 E4 M2 E4 M3 E4 ::: CLOB);                       // interleaved with 16 v_exp
    if constexpr (MODE == 4) asm volatile(E4 E4 E4 E4 ::: CLOB);                                   // 16 v_exp alone
    if constexpr (MODE == 5) asm volatile(M0 M1 M2 M3 F8 F8 F8 F8 ::: CLOB);                       // clustered: 4 MFMAs then 32 v_fma
    if constexpr (MODE == 6) asm volatile(M0 M0 M0 M0 ::: CLOB);                                   // dependent chain on one accumulator
    if constexpr (MODE == 7) asm volatile(M0 F8 E4 M1 F8 E4 M2 F8 E4 M3 F8 E4 ::: CLOB);           // 32 v_fma + 16 v_exp interleaved
    if constexpr (MODE == 8) asm volatile(F8 E4 F8 E4 F8 E4 F8 E4 ::: CLOB);                       // the same VALU without MFMAs
    if constexpr (MODE == 9) asm volatile(M0 C4 P4 D4 M1 C4 P4 D4 M2 C4 P4 D4 M3 C4 P4 D4 ::: CLOB);   // cvt_pk + pk_fma + dot2c beside MFMAs
    if constexpr (MODE == 10) asm volatile(C4 P4 D4 C4 P4 D4 C4 P4 D4 C4 P4 D4 ::: CLOB);          // the same without MFMAs
    if constexpr (MODE == 11) asm volatile(M0 M0 F8 F8 M0 M0 F8 F8 ::: CLOB);                      // dependent chain + VALU
}
template <int MA, int MB>
__global__ __launch_bounds__(1024) void probe(long *out, int iters, int split /* waves >= split run MB */) {
    const int wave = threadIdx.x >> 6;
    long t0 = 0, t1 = 0;
    if (wave < split) {
        for (int i = 0; i < 10; ++i) body<MA>();
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i) body<MA>();
        t1 = __builtin_amdgcn_s_memtime();
    } else {
        for (int i = 0; i < 10; ++i) body<MB>();
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i) body<MB>();
        t1 = __builtin_amdgcn_s_memtime();
    }
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
}
template <int MA, int MB>
static void run(const char *what, int waves, int split) {
    long *d; (void)hipMalloc(&d, 16 * 8 * 4);
    const int iters = 2000;
    hipLaunchKernelGGL((probe<MA, MB>), dim3(4), dim3(waves * 64), 0, 0, d, iters, split);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("%s: %s\n", what, hipGetErrorString(e)); fflush(stdout); return; }
    long h[64]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // s_memtime ticks at 100 MHz on gfx950?  report raw ticks per iteration for wave 0 and the first wave of the second group
    long mx = 0; for (int w = 0; w < waves; ++w) mx = h[w] > mx ? h[w] : mx;        // the slowest wave: what the SIMD needed for all of them
    printf("%-64s waves %2d: A %.1f (slowest wave %.1f)", what, waves, (double)h[0] / iters, (double)mx / iters);
    if (split < waves) printf("   B %.1f", (double)h[split] / iters);
    printf("   (ticks/iter)\n"); fflush(stdout);
    (void)hipFree(d);
}
int main() {
    run<0, 0>("4 MFMA (independent)", 4, 4);
    run<6, 6>("4 MFMA (one accumulator)", 4, 4);
    run<2, 2>("32 v_fma", 4, 4);
    run<4, 4>("16 v_exp", 4, 4);
    run<1, 1>("4 MFMA interleaved with 32 v_fma", 4, 4);
    run<5, 5>("4 MFMA then 32 v_fma (clustered)", 4, 4);
    run<3, 3>("4 MFMA interleaved with 16 v_exp", 4, 4);
    run<8, 8>("32 v_fma + 16 v_exp", 4, 4);
    run<7, 7>("4 MFMA interleaved with 32 v_fma + 16 v_exp", 4, 4);
    run<10, 10>("16 cvt_pk + 16 pk_fma + 16 dot2c", 4, 4);
    run<9, 9>("4 MFMA interleaved with 16 cvt_pk + 16 pk_fma + 16 dot2c", 4, 4);
    run<11, 11>("dependent MFMA pairs + 32 v_fma", 4, 4);
    run<12, 12>("32 v_fmamk", 4, 4);
    run<13, 13>("16 v_pk_add_f32", 4, 4);
    run<14, 14>("16 v_max3", 4, 4);
    run<15, 15>("16 v_exp_f32 (8 regs)", 4, 4);
    run<16, 16>("16 v_exp_f16", 4, 4);
    run<19, 19>("16 cvt_pk_bf16", 4, 4);
    run<20, 20>("16 dot2c (2 chains)", 4, 4);
    run<17, 17>("softmax body, 16 scores: 16 fmamk 16 exp 8 cvt 4 dot2c", 4, 4);
    run<18, 18>("softmax body, 16 scores: 8 pk_add 16 exp 8 cvt 4 dot2c", 4, 4);
    run<17, 17>("2 waves/SIMD both: softmax body (fmamk)", 8, 8);
    run<18, 18>("2 waves/SIMD both: softmax body (pk_add)", 8, 8);
    run<12, 12>("2 waves/SIMD both: 32 v_fmamk", 8, 8);
    run<15, 15>("2 waves/SIMD both: 16 v_exp_f32", 8, 8);
    run<0, 17>("A: 4 MFMA   B: softmax body", 8, 4);
    run<12, 12>("4 waves/SIMD all: 32 v_fmamk", 16, 16);
    run<15, 15>("4 waves/SIMD all: 16 v_exp_f32", 16, 16);
    run<17, 17>("4 waves/SIMD all: softmax body (fmamk)", 16, 16);
    run<18, 18>("4 waves/SIMD all: softmax body (pk_add)", 16, 16);
    run<14, 14>("4 waves/SIMD all: 16 v_max3", 16, 16);
    run<19, 19>("4 waves/SIMD all: 16 cvt_pk", 16, 16);
    run<0, 0>("4 waves/SIMD all: 4 MFMA 32x32x16", 16, 16);
    printf("-- two waves per SIMD, different mixes (A = waves 0-3, B = waves 4-7)\n");
    run<0, 2>("A: 4 MFMA   B: 32 v_fma", 8, 4);
    run<0, 4>("A: 4 MFMA   B: 16 v_exp", 8, 4);
    run<0, 8>("A: 4 MFMA   B: 32 v_fma + 16 v_exp", 8, 4);
    run<6, 8>("A: 4 MFMA one acc   B: 32 v_fma + 16 v_exp", 8, 4);
    run<0, 0>("A: 4 MFMA   B: 4 MFMA", 8, 4);
    run<2, 2>("A: 32 v_fma   B: 32 v_fma", 8, 4);
    run<7, 7>("A = B: 4 MFMA interleaved with 32 v_fma + 16 v_exp", 8, 4);
    run<5, 5>("A = B: clustered 4 MFMA then 32 v_fma", 8, 4);
    run<5, 5>("4 waves/SIMD: clustered 4 MFMA then 32 v_fma", 16, 16);
    run<7, 7>("4 waves/SIMD: 4 MFMA interleaved with 32 v_fma + 16 v_exp", 16, 16);
    return 0;
}

// mfma_ceiling.hip -- what the matrix pipe of THIS MI355X sustains under its power management, independent of any GEMM structure.
//   build: make -C tools mfma_ceiling.bin      run (GPU box): tools/mfma_ceiling.bin [ms_per_case]
// Every CU runs 8 waves (2 per SIMD, like the GEMM) of back-to-back v_mfma_f32_32x32x16 on register operands -- no LDS, no memory, no
// barriers -- for a fixed wall time.  Reported per case: TFLOP/s, the shader clock the chip actually ran at (cycle counter / 100 MHz
// wall clock) and the MFMA issue duty (MFMA cycles / elapsed cycles).  Cases: operand fill (zero / constant / uniform random in
// [-1, 1)), dtype, and a duty-cycled variant that idles the pipe part of the time (s_sleep) to show the power/clock trade.
// cdna_hip_programming.md rule 25: the same kernel runs 15-21 % faster on zero-filled operands (DVFS) -- quote the random-data row.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__device__ __forceinline__ float rnd_unit(uint32_t s) { return (float)(hash32(s) >> 8) * (1.0f / 8388608.0f) - 1.0f; }

struct Out { unsigned long long cycles, realtime, mfmas; float sink; };

// FILL: 0 zero, 1 constant 0.5, 2 random.  SLEEP: s_sleep argument after every 32 MFMAs (0 = none).
template <typename V, typename E, int FILL, int SLEEP>
__global__ __launch_bounds__(512, 2) void mfma_loop(Out *out, int iters) {
    const int tid = threadIdx.x + blockIdx.x * blockDim.x;
    V a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float va = FILL == 0 ? 0.0f : (FILL == 1 ? 0.5f : rnd_unit(tid * 64 + i * 8 + e));
            const float vb = FILL == 0 ? 0.0f : (FILL == 1 ? 0.5f : rnd_unit(tid * 64 + 32 + i * 8 + e) * 0.05f);
            a[i][e] = (E)va; b[i][e] = (E)vb;
        }
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    __syncthreads();
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (sizeof(E) == 2 && __is_same(E, __bf16)) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8 *)&a[(i + k) & 3], *(bf16x8 *)&b[i & 3], acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(half8 *)&a[(i + k) & 3], *(half8 *)&b[i & 3], acc[i], 0, 0, 0);
            }
        if constexpr (SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);
        // keep the accumulators bounded on random data without touching the pipe's work: nothing (values grow ~ sqrt(iters), far from overflow)
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][15];
    if (threadIdx.x == 0) { Out o; o.cycles = c1 - c0; o.realtime = r1 - r0; o.mfmas = (unsigned long long)iters * 32; o.sink = s; out[blockIdx.x] = o; }
}

// the same loop on v_mfma_f32_16x16x32 (4 accumulator registers per MFMA, 16 cycles each): 64 MFMAs per iteration = the same flops
typedef float f32x4 __attribute__((ext_vector_type(4)));
// ORDER 4: as 0, with the accumulators in AGPRs (inline asm; hipcc's own choice for these kernels is the VGPR form)
// ORDER: which operand registers consecutive MFMAs share (operand-bus toggling): 0 both change every MFMA, 1 the first (A) operand is
// held for 4 MFMAs, 2 the second (B) operand is held for 4, 3 both held (one register pair for everything)
template <int FILL, int ORDER = 0>
__global__ __launch_bounds__(512, 2) void mfma16_loop(Out *out, int iters) {
    const int tid = threadIdx.x + blockIdx.x * blockDim.x;
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            a[i][e] = (__bf16)(FILL == 0 ? 0.0f : rnd_unit(tid * 64 + i * 8 + e));
            b[i][e] = (__bf16)(FILL == 0 ? 0.0f : rnd_unit(tid * 64 + 32 + i * 8 + e) * 0.05f);
        }
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
    __syncthreads();
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int ia = ORDER == 0 ? (i + k) & 3 : (ORDER == 1 ? (i >> 2) & 3 : (ORDER == 2 ? i & 3 : 0));
                const int ib = ORDER == 0 ? i & 3 : (ORDER == 1 ? i & 3 : (ORDER == 2 ? (i >> 2) & 3 : 0));
                if constexpr (ORDER == 4) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a[(i + k) & 3]), "v"(b[i & 3]));
                else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ia], b[ib], acc[i], 0, 0, 0);
            }
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
    if (threadIdx.x == 0) { Out o; o.cycles = c1 - c0; o.realtime = r1 - r0; o.mfmas = (unsigned long long)iters * 64; o.sink = s; out[blockIdx.x] = o; }
}
template <int FILL, int ORDER = 0>
static void run16(const char *name, int n_cu, double target_ms) {
    Out *d; CK(hipMalloc(&d, sizeof(Out) * n_cu));
    int iters = 2000;
    for (int pass = 0; pass < 2; ++pass) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((mfma16_loop<FILL, ORDER>), dim3(n_cu), dim3(512), 0, 0, d, iters);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (pass == 0) { iters = (int)(iters * target_ms / ms); if (iters < 100) iters = 100; continue; }
        Out *h = (Out *)malloc(sizeof(Out) * n_cu); CK(hipMemcpy(h, d, sizeof(Out) * n_cu, hipMemcpyDeviceToHost));
        double cyc = 0, rt = 0; for (int i = 0; i < n_cu; ++i) { cyc += h[i].cycles; rt += h[i].realtime; }
        const double flops = 2.0 * 16 * 16 * 32 * (double)iters * 64 * 8 * n_cu;
        printf("%-34s %8.1f TFLOP/s   kernel %7.2f ms   shader clock %6.0f MHz\n", name, flops / (ms * 1e-3) / 1e12, ms, (cyc / n_cu) / ((rt / n_cu) / 100.0));
        fflush(stdout); free(h);
        CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    }
    CK(hipFree(d));
}

template <typename V, typename E, int FILL, int SLEEP>
static void run(const char *name, int n_cu, double target_ms) {
    Out *d; CK(hipMalloc(&d, sizeof(Out) * n_cu));
    int iters = 2000;
    for (int pass = 0; pass < 2; ++pass) {      // pass 0 calibrates the iteration count to the target duration
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((mfma_loop<V, E, FILL, SLEEP>), dim3(n_cu), dim3(512), 0, 0, d, iters);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (pass == 0) { iters = (int)(iters * target_ms / ms); if (iters < 100) iters = 100; continue; }
        Out *h = (Out *)malloc(sizeof(Out) * n_cu); CK(hipMemcpy(h, d, sizeof(Out) * n_cu, hipMemcpyDeviceToHost));
        double cyc = 0, rt = 0; for (int i = 0; i < n_cu; ++i) { cyc += h[i].cycles; rt += h[i].realtime; }
        cyc /= n_cu; rt /= n_cu;
        const double flops = 2.0 * 32 * 32 * 16 * (double)iters * 32 * 8 * n_cu;          // 8 waves per workgroup
        const double mhz = cyc / (rt / 100.0);                                          // s_memrealtime ticks at 100 MHz
        const double duty = (double)iters * 32 * 2 * 32.0 / cyc;                         // two waves per SIMD, 32 cycles per MFMA
        printf("%-34s %8.1f TFLOP/s   kernel %7.2f ms   shader clock %6.0f MHz   MFMA duty %5.1f %%\n", name, flops / (ms * 1e-3) / 1e12, ms, mhz, duty * 100.0);
        fflush(stdout);
        free(h);
        CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    }
    CK(hipFree(d));
}

int main(int argc, char **argv) {
    const double ms = argc > 1 ? atof(argv[1]) : 200.0;
    int n_cu = 256; (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0);
    if (const char *e = getenv("LAB_CUS")) n_cu = atoi(e);
    printf("# %d workgroups x 8 waves, ~%.0f ms per case; nominal peak %.1f TFLOP/s at 2400 MHz\n", n_cu, ms, 2.0 * 32 * 32 * 16 / 32.0 * 4 * n_cu * 2.4e9 / 1e12);
    run<bf16x8, __bf16, 0, 0>("bf16 zero operands", n_cu, ms);
    run<bf16x8, __bf16, 1, 0>("bf16 constant 0.5", n_cu, ms);
    run<bf16x8, __bf16, 2, 0>("bf16 uniform random", n_cu, ms);
    run<half8, _Float16, 2, 0>("f16  uniform random", n_cu, ms);
    run16<0>("bf16 16x16x32 zero operands", n_cu, ms);
    run16<2>("bf16 16x16x32 uniform random", n_cu, ms);
    run16<2, 1>("bf16 16x16x32 random, A held x4", n_cu, ms);
    run16<2, 2>("bf16 16x16x32 random, B held x4", n_cu, ms);
    run16<2, 3>("bf16 16x16x32 random, A and B held", n_cu, ms);
    run16<2, 4>("bf16 16x16x32 random, acc in AGPRs", n_cu, ms);
    run16<2, 0>("bf16 16x16x32 uniform random (again)", n_cu, ms);
    run<bf16x8, __bf16, 2, 1>("bf16 random, s_sleep 1 / 32 MFMA", n_cu, ms);
    run<bf16x8, __bf16, 2, 4>("bf16 random, s_sleep 4 / 32 MFMA", n_cu, ms);
    run<bf16x8, __bf16, 2, 8>("bf16 random, s_sleep 8 / 32 MFMA", n_cu, ms);
    return 0;
}

// gemm_lab.hip -- standalone GEMM laboratory (no Python / torch): times and checks the wide-tile kernels on random data.
//   build:  make -C tools gemm_lab.bin        run (on the GPU box):  tools/gemm_lab.bin [iters] [filter]
// Every line: shape, kernel, epilogue, dtype, mean and best microseconds over `iters` launches, TFLOP/s of the mean,
// and the result of a sampled check against an f64 reference computed by a plain kernel.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../vit.cpp_amd/csrc/kernels.h"

using namespace vitx;
static const Tuning *g_tune = nullptr;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
// uniform [-scale, scale) operand values (full-range random data: cdna_hip_programming.md rule 25)
template <typename T> __global__ void fill_kernel(T *p, size_t n, uint32_t seed, float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = (T)(((float)(hash32((uint32_t)i * 2654435761U + seed) >> 8) * (1.0f / 8388608.0f) - 1.0f) * scale);
}
__global__ void fill_f32(float *p, size_t n, uint32_t seed, float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = ((float)(hash32((uint32_t)i * 2654435761U + seed) >> 8) * (1.0f / 8388608.0f) - 1.0f) * scale;
}
// reference of S sampled outputs: one thread per sample, f64 accumulation
template <typename T> __global__ void ref_kernel(const T *A, const T *W, const float *bias, int M, int N, int K, const int *sm, const int *sn, double *out, int S) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const T *a = A + (size_t)sm[s] * K, *w = W + (size_t)sn[s] * K;
    double acc = 0.0;
    for (int k = 0; k < K; ++k) acc += (double)(float)a[k] * (double)(float)w[k];
    out[s] = acc + (double)bias[sn[s]];
}
template <typename T> __global__ void gather_kernel(const void *out, int epi, int ldo, const int *sm, const int *sn, float *got, int S) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const size_t idx = (size_t)sm[s] * ldo + sn[s];
    got[s] = (epi == EPI_BIAS || epi == EPI_BIAS_GELU) ? (float)((const T *)out)[idx] : ((const float *)out)[idx];
}

static double gelu_ref(double x) { return 0.5 * x * (1.0 + tanh(0.79788456080286535588 * x * (1.0 + 0.044715 * x * x))); }

struct Shape { const char *name; int M, N, K; int Mr = 0; };   // Mr: rows stored (0 = all M): the last row block is then an edge tile
struct Variant { const char *name; int kind; int cfg; };   // kind 0: ring cfg, 1: ping-pong, 2 / 3: free-running kernels with 4 / 8 waves (gemm_w4.hip)

template <typename T>
static void run_one(const Shape &sh, const Variant &v, int epi, int dtype, int iters, int n_cu, bool check) {
    GemmArgs g{};
    const int M = sh.M, N = sh.N, K = sh.K;
    const size_t out_elem = (epi == EPI_BIAS || epi == EPI_BIAS_GELU) ? 2 : 4;
    T *A, *W; float *bias; void *out;
    CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&W, (size_t)N * K * 2)); CK(hipMalloc(&bias, (size_t)N * 4)); CK(hipMalloc(&out, (size_t)M * N * out_elem));
    fill_kernel<T><<<2048, 256>>>(A, (size_t)M * K, 1u, 1.0f);
    fill_kernel<T><<<2048, 256>>>(W, (size_t)N * K, 2u, 1.0f / sqrtf((float)K) * 2.0f);
    fill_f32<<<64, 256>>>(bias, (size_t)N, 3u, 0.5f);
    if (epi == EPI_BIAS_RESID) fill_f32<<<2048, 256>>>((float *)out, (size_t)M * N, 4u, 1.0f);
    else CK(hipMemset(out, 0, (size_t)M * N * out_elem));
    CK(hipDeviceSynchronize());
    g.A = A; g.W = W; g.bias = bias; g.out = out; g.pos = nullptr; g.M = M; g.M_real = sh.Mr > 0 ? sh.Mr : M; g.N = N; g.N_pad = N; g.K = K; g.lda = K; g.ldw = K; g.ldo = N; g.tpi = 0;
    if (const char *e = getenv("LAB_GROUP_M")) g.group_m = atoi(e);
    unsigned *tl = nullptr;
    if (v.kind == 1 && (v.cfg & 32)) { CK(hipMalloc(&tl, 256 * 8 * 64 * 4)); CK(hipMemset(tl, 0, 256 * 8 * 64 * 4)); g.pos = (const float *)tl; }
    long long *clk = nullptr;      // clock probe of the free-running kernels (and pp_clock): per workgroup {shader cycles, 100 MHz ticks, tiles, K-tiles per tile}
    if (v.kind >= 2 || (v.kind == 1 && v.cfg == 4096)) { CK(hipMalloc(&clk, 256 * 4 * 8)); CK(hipMemset(clk, 0, 256 * 4 * 8)); g.pos = (const float *)clk; }
    if (v.kind == 1 && (v.cfg & (28 | 2048))) check = false;
    if (v.kind == 1 && v.cfg == 4096) check = true;
    const bool brief = false;
    if (v.kind >= 2 && (v.cfg & (28 | 64 | 2048))) check = false;
    auto launch = [&]() -> hipError_t { return v.kind == 0 ? launch_gemm_ring(*g_tune, dtype, epi, g, v.cfg, 0) : v.kind >= 2 ? launch_gemm_w4(dtype, epi, g, n_cu, 0, v.cfg, false, v.kind == 2 ? 4 : 8) : launch_gemm_pp(dtype, epi, g, n_cu, 0, v.cfg); };

    // ---- check first (on a fresh output buffer): 4096 sampled outputs incl. the corners of the first and last tile
    char verdict[96] = "unchecked";
    if (check) {
        const int S = 4096;
        std::vector<int> sm(S), sn(S);
        uint32_t r = 12345u;
        const int Mr = g.M_real, npad = Mr < M ? 256 : 0;        // the last `npad` samples probe the pad rows Mr .. M: they must keep their fill (0)
        for (int s = 0; s < S; ++s) { r = r * 1664525u + 1013904223u; sm[s] = (r >> 8) % Mr; r = r * 1664525u + 1013904223u; sn[s] = (r >> 8) % N; }
        sm[0] = 0; sn[0] = 0; sm[1] = Mr - 1; sn[1] = N - 1; sm[2] = 0; sn[2] = N - 1; sm[3] = Mr - 1; sn[3] = 0; sm[4] = 255; sn[4] = 255; sm[5] = 128; sn[5] = 64;
        for (int s = S - npad; s < S; ++s) sm[s] = Mr + (s * 7) % (M - Mr);
        int *dsm, *dsn; double *dref; float *dgot, *dprev;
        CK(hipMalloc(&dsm, S * 4)); CK(hipMalloc(&dsn, S * 4)); CK(hipMalloc(&dref, S * 8)); CK(hipMalloc(&dgot, S * 4)); CK(hipMalloc(&dprev, S * 4));
        CK(hipMemcpy(dsm, sm.data(), S * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dsn, sn.data(), S * 4, hipMemcpyHostToDevice));
        ref_kernel<T><<<(S + 63) / 64, 64>>>(A, W, bias, M, N, K, dsm, dsn, dref, S);
        if (out_elem == 4) gather_kernel<T><<<(S + 63) / 64, 64>>>(out, EPI_BIAS_F32, N, dsm, dsn, dprev, S);       // residual values before the launch
        else CK(hipMemset(dprev, 0, S * 4));
        hipError_t le = launch();
        if (le != hipSuccess) { snprintf(verdict, sizeof verdict, "LAUNCH FAILED: %s", hipGetErrorString(le)); }
        else {
            gather_kernel<T><<<(S + 63) / 64, 64>>>(out, epi, N, dsm, dsn, dgot, S);
            CK(hipDeviceSynchronize());
            std::vector<double> ref(S); std::vector<float> got(S), prev(S);
            CK(hipMemcpy(ref.data(), dref, S * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(got.data(), dgot, S * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(prev.data(), dprev, S * 4, hipMemcpyDeviceToHost));
            double worst = 0.0; int bad = 0;
            const double ulp = dtype == DT_F16 ? 1.0 / 1024 : 1.0 / 128;
            for (int s = 0; s < S; ++s) {
                double want = ref[s], tol = 2e-4 * (1.0 + fabs(want));
                if (epi == EPI_BIAS) tol += ulp * fabs(want);
                if (epi == EPI_BIAS_GELU) { want = gelu_ref(want); tol = 2.5 * ulp * (fabs(want) + 0.02) + 3e-3 * ulp * 128; }
                if (epi == EPI_BIAS_RESID) want += (double)prev[s];
                if (s >= S - npad) { want = (double)prev[s]; tol = 0.0; if ((double)got[s] == want) continue; }
                const double err = fabs((double)got[s] - want) / tol;
                if (!(err <= 1.0)) { if (bad < 16 && getenv("LAB_SHOWBAD")) printf("   bad: m %d (%d) n %d (%d) got %g want %g\n", sm[s], sm[s] % 256, sn[s], sn[s] % 256, got[s], want); ++bad; }
                if (!(err <= worst)) worst = err;
            }
            if (bad) snprintf(verdict, sizeof verdict, "CHECK FAILED %d/%d worst %.1f tol", bad, S, worst);
            else snprintf(verdict, sizeof verdict, "ok (worst %.2f tol)", worst);
            // the free-running kernels claim the K order of every other family: the WHOLE output must equal the ping-pong kernel's bit for bit
            if (!bad && v.kind >= 2 && v.cfg == 0 && out_elem == 2 && gemm_pp_supports(g)) {
                const size_t nb = (size_t)M * N * out_elem;
                void *out2; CK(hipMalloc(&out2, nb)); CK(hipMemset(out2, 0, nb));
                GemmArgs g2 = g; g2.out = out2; g2.pos = nullptr;
                if (launch_gemm_pp(dtype, epi, g2, n_cu, 0, 0) == hipSuccess) {
                    CK(hipDeviceSynchronize());
                    std::vector<uint16_t> a(nb / 2), b(nb / 2);
                    CK(hipMemcpy(a.data(), out, nb, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), out2, nb, hipMemcpyDeviceToHost));
                    size_t diff = 0; for (size_t i = 0; i < a.size(); ++i) diff += a[i] != b[i];
                    const size_t l = strlen(verdict);
                    if (diff) snprintf(verdict + l, sizeof verdict - l, "  BITS DIFFER from pp in %zu outputs", diff); else snprintf(verdict + l, sizeof verdict - l, "  == pp bit for bit");
                }
                CK(hipFree(out2));
            }
        }
        (void)hipFree(dsm); (void)hipFree(dsn); (void)hipFree(dref); (void)hipFree(dgot); (void)hipFree(dprev);
    }
    // ---- timing: 2 warm-up launches, then `iters` launches timed one by one
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) (void)launch();
    CK(hipDeviceSynchronize());
    std::vector<float> t(iters);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) (void)launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float total = 0; CK(hipEventElapsedTime(&total, e0, e1));
    float best = 1e30f;
    for (int i = 0; i < std::min(iters, 5); ++i) {
        CK(hipEventRecord(e0, 0)); (void)launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
    }
    const double us = total / iters * 1e3, tf = 2.0 * M * N * (double)K / (total / iters) / 1e9;
    printf("%-7s M=%-6d N=%-5d K=%-5d %-9s epi=%d %s  mean %9.1f us  best %9.1f us  %7.1f TF/s  %s\n", sh.name, M, N, K, v.name, epi, dtype == DT_F16 ? "f16 " : "bf16",
           us, best * 1e3, tf, verdict);
    fflush(stdout);
    if (clk) {
        std::vector<long long> h(256 * 4);
        CK(hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost));
        double cyc = 0, tick = 0, kt = 0; int nb = 0;
        for (int b = 0; b < 256; ++b) if (h[b * 4 + 2] > 0) { cyc += (double)h[b * 4]; tick += (double)h[b * 4 + 1]; kt += (double)h[b * 4 + 2] * h[b * 4 + 3]; ++nb; }
        if (nb) printf("   clock: %d workgroups, mean %.0f shader cycles in %.2f us = %.0f MHz effective; %.0f cycles per K-tile (epilogues included)\n", nb, cyc / nb, tick / nb / 100.0, cyc / tick * 100.0, cyc / kt);
        CK(hipFree(clk));
    }
    if (tl) {      // timeline: stamps after each barrier (2 per phase, 16 phases = K-tiles 4..7 of the first tile), waves 0 and 4 of a few workgroups
        std::vector<unsigned> h(256 * 8 * 64);
        CK(hipMemcpy(h.data(), tl, h.size() * 4, hipMemcpyDeviceToHost));
        for (int b : {0, 1, 100, 255}) for (int w : {0, 4}) {
            const unsigned *st = &h[((size_t)b * 8 + w) * 64];
            printf("   timeline block %3d wave %d: deltas:", b, w);
            const int ns = 32;          // two stamps per phase, eight per K-tile
            for (int i = 1; i < ns; ++i) printf(i % 8 == 0 ? " | %u" : " %u", st[i] - st[i - 1]);
            printf("\n");
        }
        if (v.cfg & 1024) {     // arrival stamps: per barrier, arrival of the first-row wave 0 and of the second-row wave 4 (which is one barrier behind)
            const unsigned *w0 = &h[0], *w4 = &h[4 * 64];
            printf("   barrier#: release(~max arrival) interval | w0 slack, w4 slack  (w4 stamp i belongs to barrier i+1)\n");
            unsigned prev = 0;
            for (int i = 1; i < (brief ? 16 : 31); ++i) {
                const unsigned a0 = w0[i], a4 = w4[i - 1];       // both arrive at global barrier i
                const unsigned rel = (int)(a0 - a4) > 0 ? a0 : a4;
                printf("   b%02d %s: interval %4u  w0 slack %4d  w4 slack %4d\n", i, (i & 1) ? "w0:B2/w4:B1" : "w0:B1/w4:B2", prev ? rel - prev : 0, (int)(rel - a0), (int)(rel - a4));
                prev = rel;
            }
        }
        const unsigned *a0 = &h[0], *a4 = &h[4 * 64];
        printf("   wave4 - wave0 stamp offsets:"); for (int i = 0; i < 32; ++i) printf(" %d", (int)(a4[i] - a0[i])); printf("\n");
        CK(hipFree(tl));
    }
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(bias)); CK(hipFree(out));
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 10;
    const char *filter = argc > 2 ? argv[2] : "";
    int n_cu = 256; (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0);
    if (const char *e = getenv("LAB_CUS")) n_cu = atoi(e);
    g_tune = tuning_for_device(0);
    if (!g_tune) { fprintf(stderr, "kernel bring-up failed\n"); return 2; }
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("# device %s, %d CUs, clock %d MHz; iters %d\n", prop.name, prop.multiProcessorCount, prop.clockRate / 1000, iters);
    const Shape shapes[] = {
        {"tiny", 512, 512, 256}, {"edge", 768, 1024, 384}, {"sq4k", 4096, 4096, 4096}, {"sq8k", 8192, 8192, 8192},
        {"qkv", 50432, 2304, 768}, {"proj", 50432, 768, 768}, {"fc1", 50432, 3072, 768}, {"fc2", 50432, 768, 3072},
        {"hqkv", 25344, 2304, 768, 25216}, {"hfc1", 25344, 3072, 768, 25216},      // an average sub-batch of the forward (128 images; the last row block is an edge tile)
        {"ragged", 1280, 768, 512, 1100}, {"k256", 1024, 512, 256}, {"k128", 512, 512, 128},
        {"qkvL", 73984, 3072, 1024}, {"fc2L", 73984, 1024, 4096},
    };
    // kind 1 = ping-pong kernel, cfg = its FLAGS (gemm_pp.hip; non-zero builds exist under -DVITX_LAB only, which this tool is compiled with)
    const Variant variants[] = {{"ring945", 0, 945}, {"pp", 1, 0},  {"pp_noprio", 1, 1}, {"pp_nodma", 1, 4}, {"pp_noread", 1, 8}, {"pp_mfmaonly", 1, 12},
                                {"pp_nomfma", 1, 16}, {"pp_stamp", 1, 32}, {"pp_drain", 1, 512}, {"pp_noepi", 1, 2048}, {"pp_clock", 1, 4096},
                                {"w4", 2, 0}, {"w4_nodma", 2, 4}, {"w4_noread", 2, 8}, {"w4_mfmaonly", 2, 12}, {"w4_nomfma", 2, 16}, {"w4_nobar", 2, 64}, {"w4_bare", 2, 76}, {"w4_noepi", 2, 2048},
                                {"w8", 3, 0}, {"w8_nodma", 3, 4}, {"w8_noread", 3, 8}, {"w8_mfmaonly", 3, 12}, {"w8_nomfma", 3, 16}, {"w8_nobar", 3, 64}, {"w8_bare", 3, 76}, {"w8_noepi", 3, 2048}};
    for (const Shape &sh : shapes)
        for (const Variant &v : variants) {
            int epis[4] = {EPI_BIAS, -1, -1, -1};
            if (!strcmp(sh.name, "proj") || !strncmp(sh.name, "fc2", 3)) epis[0] = EPI_BIAS_RESID;
            if (!strcmp(sh.name, "fc1") || !strcmp(sh.name, "hfc1")) epis[0] = EPI_BIAS_GELU;
            if (!strcmp(sh.name, "tiny") || !strcmp(sh.name, "edge")) { epis[1] = EPI_BIAS_GELU; epis[2] = EPI_BIAS_RESID; epis[3] = EPI_BIAS_F32; }
            if (!strcmp(sh.name, "ragged") || !strcmp(sh.name, "k256") || !strcmp(sh.name, "k128")) epis[1] = EPI_BIAS_GELU;
            for (int e = 0; e < 4; ++e) {
                if (epis[e] < 0 || (v.kind >= 1 && v.cfg && epis[e] != EPI_BIAS)) continue;
                if (v.kind >= 2 && epis[e] != EPI_BIAS && epis[e] != EPI_BIAS_GELU) continue;
                if (v.kind == 0 && sh.Mr) continue;
                for (int dtype = 0; dtype < 2; ++dtype) {
                    if (dtype == 0 && strcmp(sh.name, "tiny") && strcmp(sh.name, "edge") && strcmp(sh.name, "qkv") && strcmp(sh.name, "ragged") && strcmp(sh.name, "k256")) continue;    // f16: correctness shapes + one big one
                    char tag[96]; snprintf(tag, sizeof tag, "%s:%s:%d:%s", sh.name, v.name, epis[e], dtype ? "bf16" : "f16");
                    if (filter[0]) {       // comma-separated substrings: any match runs the case
                        bool hit = false; std::string f(filter); size_t pos = 0;
                        while (pos <= f.size()) { size_t c = f.find(',', pos); if (c == std::string::npos) c = f.size(); if (c > pos && strstr(tag, f.substr(pos, c - pos).c_str())) hit = true; pos = c + 1; }
                        if (!hit) continue;
                    }
                    const bool small = sh.M <= 1280;
                    if (dtype == DT_F16) run_one<_Float16>(sh, v, epis[e], dtype, small ? 3 : iters, n_cu, true);
                    else run_one<__bf16>(sh, v, epis[e], dtype, small ? 3 : iters, n_cu, true);
                }
            }
        }
    return 0;
}

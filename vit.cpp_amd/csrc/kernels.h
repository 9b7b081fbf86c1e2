// kernels.h -- launchers of the hand-written gfx950 kernels (kernels.hip).
// All pointers are device pointers; every launcher only enqueues on `stream`.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vitx {

enum { DT_F16 = 0, DT_BF16 = 1 };

// GEMM epilogues (C = A[M][K] . W[N][K]^T, f32 accumulate)
enum {
    EPI_BIAS = 0,        // out(dtype)[m][n] = acc + bias[n]                       qkv      (vit.cpp:820-821)
    EPI_BIAS_GELU = 1,   // out(dtype)[m][n] = gelu_tanh(round(acc + bias[n]))     fc1      (vit.cpp:889-893)
    EPI_BIAS_RESID = 2,  // out(f32)[m][n]   = (acc + bias[n]) + out[m][n]         proj/fc2 (vit.cpp:868-873, 896-900)
    EPI_BIAS_F32 = 3,    // out(f32)[m][n]   = acc + bias[n]                       head     (vit.cpp:927-928)
    EPI_PATCH = 4,       // out(f32)[m + m/tpi + 1][n] = (acc + bias[n]) + pos[(m%tpi + 1)][n]   (vit.cpp:772-797)
    EPI_BIAS_HILO = 5    // qkv of the F16 parity mode: v = acc + bias[n] kept to f32 grade as TWO 16-bit planes, out[m][n] = hi = round(v) and
                         // out[hilo_off + ..] = lo = round((v - hi) * 2048): the reference's q, k, v stay f32 into the attention products
                         // (vit.cpp:826-858: ggml_mul_mat of f32 views), and hi + lo / 2048 reproduces v to 2^-22
};
constexpr float kHiLoScale = 2048.0f, kHiLoInv = 1.0f / 2048.0f;

struct GemmArgs {
    const void *A; const void *W; const float *bias; void *out; const float *pos;
    int M;        // rows computed (multiple of the M tile; buffers are padded to it)
    int M_real;   // rows stored
    int N;        // columns stored (real)
    int N_pad;    // columns of W available (multiple of the N tile; zero rows beyond N)
    int K;        // multiple of 64
    int lda, ldw, ldo;
    int tpi;      // EPI_PATCH: patch tokens per image (g*g)
    long hilo_off;  // EPI_BIAS_HILO: ELEMENT offset of the lo plane behind `out` (a whole number of rows; the byte offset must fit 32 bits)
    int dbg;      // ablation bits of the ring kernel's laboratory build (VITX_LAB only; 0 in the product)
    // q4_0 weights kept in block form (launch_gemm_q4 only): W = nibble plane [N_pad][K/2] bytes (16 per block), Wscale = f16 block
    // scales [N_pad][K/32]; both planes are the file's block_q4_0 fields re-laid out, 4.5 bits per weight
    const uint16_t *Wscale;
    int group_m;  // ping-pong kernel: m-tiles per raster group (0 = the default, 8)
    // LayerNorm of the output rows fused into an EPI_BIAS_RESID GEMM on the ping-pong kernel (gemm_ln_fusable()): `ln` != nullptr.
    const struct GemmLn *ln;
    // A GEMM whose A operand is the output of a LayerNorm-fusing GEMM: `fix` = that launch's GemmLn (+ x = its X).  Before it loads anything,
    // every workgroup normalises the row blocks IT is going to read that were left behind (todo[rb] == epoch) -- no extra launch, no
    // dependency between workgroups (several may redo the same block: identical bits).  gemm_fix_capable() says whether the kernel the
    // shape selects does this; otherwise the caller launches launch_layernorm_fixup.
    const struct GemmLn *fix;
};

// The LayerNorm that follows a residual GEMM (vit.cpp:881-885 after proj, :808-812 of the next layer after fc2), computed by the
// GEMM's own epilogue: every workgroup reduces its 256-column tile's per-row statistics from the accumulators, publishes them as
// data-tagged 8-byte granules, reads the statistics of the row block's other column tiles from its raster-adjacent peers (same
// round, same XCD), and normalises ITS OWN tile from registers into `out`: X is not re-read, no extra launch, no single-CU tail.
// All M (padded) rows are computed and stored: the buffers are padded to the row tile, pad rows stay finite and are never read
// for real rows.  A peer that does not answer within `timeout` (two such GEMMs on two streams can each hold CUs the other's
// workgroups wait for) makes the workgroup skip its tile and set todo[row block] = epoch; launch_layernorm_fixup() then
// normalises exactly those row blocks from X with the stand-alone arithmetic -- the same bits (device_common.h "LayerNorm statistics").
struct GemmLn {
    const float *w, *b;           // [N]
    const float *x;               // consumer-side fix only (GemmArgs::fix): the f32 rows [M][N] the statistics are recomputed from
    void *out;                    // [M][N] operand type
    float eps;
    unsigned long long *sync;     // [M / 256][N / 256][256][2] granules {value bits, tag = epoch}, zeroed once
    unsigned *todo;               // [M / 256], zeroed once
    unsigned *fallbacks;          // [1] diagnostic counter: tiles that took the fix-up path
    unsigned epoch;               // unique per launch within the process, never 0 (frozen under graph replay: callers do not fuse while capturing)
    unsigned timeout;             // ticks of the 100 MHz wall clock
    int test;                     // parity tests only: 1 = every 5th tile pretends a time-out, 3 = and does not publish (its peers really time out)
};

// ---- block-quantised weights resident in HBM (quant.hip) -------------------------------------------
// ggml block types of the reference's quantised files (vit.cpp:384-414 picks the type, quantize.cpp:271-303 writes it)
enum { QT_Q4_0 = 2, QT_Q4_1 = 3, QT_Q5_0 = 6, QT_Q5_1 = 7, QT_Q8_0 = 8 };
// One matrix to expand: `src` holds N rows of nbk blocks in the file's layout (q4_0: the two planes described at GemmArgs::Wscale,
// scales at `scales`), dst is [n_pad][nbk * 32] in the operand type; rows N..n_pad are written as zeros.
struct DequantJob { const void *src; const void *scales; void *dst; int N, n_pad, nbk; };
// Expands up to 4 matrices of one block type in ONE launch: value = exactly what HostTensor::decode_f32 computes, rounded once (RNE) to
// the operand type -- bit-identical to the host-side expansion at upload.
hipError_t launch_dequant(int dtype, int qtype, const DequantJob *jobs, int njobs, hipStream_t stream);
// C = A . dequant(W)^T with the q4_0 blocks expanded inside the GEMM's LDS-fill path (128x128x64 tiles; any epilogue)
hipError_t launch_gemm_q4(int dtype, int epi, const GemmArgs &a, hipStream_t stream);
bool gemm_q4_supports(const GemmArgs &a);

// Per-device launch parameters: one immutable copy per device, built by tuning_for_device() under a lock, so contexts on several GPUs
// (or host threads) of one process never share launch state.  The product library reads NO environment variable here: the family
// overrides below are explicit parameters of vitx_op_gemm_ex / vitx_op_attention_ex (parity tests); only the laboratory build
// (-DVITX_LAB, tools/) maps VITX_* variables onto them.
enum { ATTN_AUTO = 0, ATTN_SINGLE = 1, ATTN_FLOW = 3, ATTN_PERSIST = 4, ATTN_STREAM = 5 };
struct Tuning {
    int device = 0;
    int n_cu = 256;          // compute units of THIS device
    int n_xcd = 8;           // XCDs (hipDeviceAttributeNumberOfXccs): the LayerNorm-fusing GEMM's peer mapping is built for exactly 8
    int gemm_cfg = -1;       // -1 automatic, 1 = the ping-pong kernel forced, else a ring configuration (445, 945, 245, 122)
    int skinny_tiles = 128;  // 64x128 tiles (cfg 122) when fewer than this many 128x256 tiles exist (r02f: 64 -> 128, batches of 4-16 images)
    int gemm_split = 0;      // 1: tail rows of a partial round re-tiled 128x256 in a second launch
    int gemm_balance = 1;    // 0: one workgroup per CU even when the last round of tiles is partial
    int group_m = 0;         // raster group height of the ping-pong kernel (0 = its default)
    int pp_flags = 0;        // ablation build of the ping-pong kernel (exists under VITX_LAB only)
    int gemm_dbg = 0;        // ablation bits of the ring kernel (honoured under VITX_LAB only)
    int attn_kernel = ATTN_AUTO;
    int attn_flags = 0;      // ablation build of the pipelined attention kernel (exists under VITX_LAB only)
    int attn_grid = 0;       // persistent attention: workgroups (0 = one per CU)
};
// Looks the device up (hipGetDevice when device < 0) and on first use of a device
// sets the dynamic-LDS attribute of every kernel instantiation on it.  Thread-safe.  Returns nullptr if HIP fails.
const Tuning *tuning_for_device(int device);

hipError_t launch_gemm(const Tuning &t, int dtype, int epi, const GemmArgs &a, hipStream_t stream);
// true when launch_gemm runs this EPI_BIAS_RESID GEMM on the ping-pong kernel in one launch with whole rows (N == N_pad == ldo,
// N / 256 <= 4 column tiles), so that GemmArgs::ln may be set; the caller then launches launch_layernorm_fixup instead of launch_layernorm
bool gemm_ln_fusable(const Tuning &t, const GemmArgs &a);
// true when launch_gemm runs this GEMM on the ping-pong kernel, which honours GemmArgs::fix (K = the LayerNorm's row length, 256 .. 1024)
bool gemm_fix_capable(const Tuning &t, const GemmArgs &a);
// persistent grid of that GEMM (for the sub-batch cost model): workgroups per XCD are a multiple of the column tiles
int gemm_ln_grid(int n_cu, int M, int N);
// normalises the row blocks a fused GEMM left behind (todo[rb] == epoch) from x; a few microseconds when there are none
hipError_t launch_layernorm_fixup(int dtype, const float *x, const float *w, const float *b, void *y, int M, int D, float eps, const unsigned *todo, unsigned epoch, hipStream_t stream);
int gemm_tile_m();   // M granularity the GEMM needs (buffer row padding)
int gemm_tile_n();

// Patch embedding in one launch (patch_embed.hip; vit.cpp:747-797): X[b * N + 1 + t][:] = W . patch(b, t) + bias + pos[1 + t], X[b * N][:] = cls + pos[0].
// img: f32 HWC [n_img][S][S][Cin] (Cin = 3: RGB classifier input; 1: the grey ViTSTR input, extensions/vitstr.cpp/vitstr.cpp:713-731);
// w_perm: the [n_pad][k_pad] operand-type kernel with its K axis permuted by patch_embed_permute_k (host side, at upload); pos [N][D], cls [D].
hipError_t launch_patch_embed(int dtype, const float *img, const void *w_perm, const float *bias, const float *pos, const float *cls, float *X,
                              int n_img, int S, int P, int Cin, int D, int n_pad, int k_pad, hipStream_t stream, bool prepare = false);
void patch_embed_permute_k(const uint16_t *w, uint16_t *w_perm, int N, int Cin, int P, int k_pad);
// y[r][:] (dtype) = LN(x[r*ldx ...]) * w + b   (vit.cpp:808-812)
// group > 1: input row r = x + (r / group) * gstride + (r % group) * ldx (the first `group` tokens of every image: ViTSTR head)
hipError_t launch_layernorm(int dtype, const float *x, long ldx, const float *w, const float *b, void *y, long ldy, int M, int D, float eps, hipStream_t stream, int group = 1, long gstride = 0);
// fused per-(image,head) attention  (vit.cpp:826-866)
hipError_t launch_attention(const Tuning &t, int dtype, const void *qkv, void *out, int n_img, int N, int D, int H, hipStream_t stream);
// Streaming two-pass kernel (attention_stream.hip), head dim 64, any token count.  precise = false: the long-sequence kernel of both
// operand types; precise = true (f16 only): the F16 parity mode's f32-grade products -- qkv is then the HI plane of the QKV GEMM's
// EPI_BIAS_HILO output and the LO plane lies lo_off elements behind it.  n_img == 0: device bring-up (dynamic-LDS attribute).
hipError_t launch_attention_stream(int dtype, bool precise, const void *qkv, void *out, int n_img, int N, int D, int H, long lo_off, hipStream_t stream);
bool attention_stream_supports(int n_img, int N, int D, int H);
// x[n] f32 -> hi[n] = round(x), lo[n] = round((x - hi) * 2048) in the operand type (what EPI_BIAS_HILO emits; parity-test entry point)
hipError_t launch_split_hilo(int dtype, const float *x, void *hi, void *lo, size_t n, hipStream_t stream);
// Attention of token 0 only (the last layer of a classifier needs nothing else, vit.cpp:910-911): out[b][D] (dtype) from qkv[n_img * N][3 D]
// (lo_off != 0: the parity mode's lo plane, F16 only); xc != nullptr: also xc[b][D] = x[b * N][D] (the class rows of the f32 residual stream)
hipError_t launch_attention_cls(int dtype, const void *qkv, long lo_off, void *out, const float *x, float *xc, int n_img, int N, int D, int H, hipStream_t stream);
bool attention_cls_supports(int N, int D, int H);  // head_dim 8, 16, 32, 64 or 128
bool attention_supports(int N, int D, int H);     // any token count; head_dim 64 (tuned kernels) or any other multiple of 8 up to 128 (generic kernel)
bool attention_single_pass_supports(int N);       // instantiation table of the register-resident kernel
bool layernorm_supports(int D);
// class softmax with the reference's fp16 (or bf16) exp rounding (vit.cpp:931)
hipError_t launch_softmax(int dtype, const float *logits, float *probs, int rows, int cols, int ld, hipStream_t stream);
hipError_t launch_preprocess(const void *u8, float *out, int n, int nx, int ny, int S, int bicubic, hipStream_t stream);
// out[row][k] = {f32 probability, i32 class} of the k largest entries of probs[row][0..cols), descending, ties by the lower class index
// (the sort of vit_predict, vit.cpp:1043-1057, on the device: the multi-GPU gather then moves 8 k bytes per row instead of 4 cols)
hipError_t launch_topk(const float *probs, int rows, int cols, int k, void *out_pairs, hipStream_t stream);
// one workgroup that does nothing for `microseconds` of the 100 MHz wall clock (stream-concurrency probe of the execution context)
hipError_t launch_spin(int microseconds, hipStream_t stream);
// the same, writing its first and last wall-clock reading (100 MHz ticks) to stamps[0..1] (device memory)
hipError_t launch_spin_stamp(int microseconds, long long *stamps, hipStream_t stream);


// internal: kernel families.  `prepare` = only set the dynamic-LDS attribute of the instantiation (device bring-up).
hipError_t launch_gemm_ring(const Tuning &t, int dtype, int epi, const GemmArgs &a, int cfg, hipStream_t stream, bool prepare = false);
bool gemm_ring_supports(const GemmArgs &a, int cfg);
hipError_t launch_gemm_pp(int dtype, int epi, const GemmArgs &a, int n_cu, hipStream_t stream, int flags = 0, bool prepare = false);
bool gemm_pp_supports(const GemmArgs &a);
// free-running wide kernels (gemm_w4.hip; LABORATORY BUILD ONLY: tools/Makefile), waves = 8: two waves per SIMD, 128 x 64 of C per wave; 4: one
// per SIMD, 128 x 128 per wave.  EPI_BIAS, EPI_BIAS_GELU; same bits as the other families
hipError_t launch_gemm_w4(int dtype, int epi, const GemmArgs &a, int n_cu, hipStream_t stream, int flags = 0, bool prepare = false, int waves = 8);
bool gemm_w4_supports(const GemmArgs &a);

}  // namespace vitx

// engine.cpp -- execution context and forward pass of the MI355X ViT engine.
//
// Replaces vit_state + vit_encode_image + the compute half of vit_predict
// (/root/reference/vit.cpp:718-941, 1004-1040).  Differences by design: the batch is
// n images (the reference hard-wires 1, vit.cpp:747), weights live in HBM in the MFMA
// operand type, all activation scratch is allocated once per context (the reference
// builds the graph twice and reallocates per call, vit.cpp:1009-1035), and the ~90
// launches of a forward are enqueued on one HIP stream without host synchronisation.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <tuple>
#include <vector>

#include "kernels.h"
#include "model_file.h"

using namespace vitx;

#define HIP_TRY(expr)                                                                                      \
    do {                                                                                                   \
        hipError_t e__ = (expr);                                                                           \
        if (e__ != hipSuccess) { set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); return VITX_ERR_HIP; } \
    } while (0)

namespace {

enum ProfClass {
    PC_GEMM_PATCH = 0, PC_LAYERNORM, PC_GEMM_QKV, PC_ATTENTION, PC_GEMM_PROJ, PC_GEMM_FC1, PC_GEMM_FC2,
    PC_GEMM_HEAD, PC_SOFTMAX, PC_DEQUANT, PC_ATTENTION_CLS, PC_GEMM_TAIL, PC_COUNT
};
const char *kProfNames[PC_COUNT] = {"patch_embed", "layernorm", "gemm_qkv_bias", "attention", "gemm_proj_resid",
                                    "gemm_fc1_gelu", "gemm_fc2_resid", "gemm_head", "softmax", "dequant_weights", "attention_cls", "gemm_cls_tail"};

// A weight matrix kept in the file's block form on the device (quant.hip): `blocks` = N rows of K/32 blocks in the file's byte
// layout -- except q4_0, which is split into a nibble plane (`blocks`, 16 B per block, rows padded to n_pad) and an f16 scale
// plane (`scales`) so both the dequant kernel and the fused small-batch GEMM read aligned 16-byte pieces.  Same bits, same size.
struct QuantW {
    void *blocks = nullptr; uint16_t *scales = nullptr;
    int type = 0, N = 0, K = 0, n_pad = 0;
};
enum { W_QKV = 0, W_PROJ, W_FC1, W_FC2, W_PER_LAYER };
struct LayerW {
    float *ln1_w, *ln1_b, *ln2_w, *ln2_b, *qkv_b, *proj_b, *fc1_b, *fc2_b;
    void *qkv_w, *proj_w, *fc1_w, *fc2_w;      // expanded operand-type matrices; nullptr where the blocks stay quantised (q[])
    QuantW q[W_PER_LAYER];
};

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

}  // namespace

struct vitx_ctx {
    const vitx_model *model = nullptr;
    vitx_hparams hp{};
    int device = 0, dtype = VITX_F16, max_batch = 0;
    int D = 0, L = 0, H = 0, C = 0, P = 0, S = 0, g = 0, N = 0, Kpe = 0, Kpe_pad = 0, C_pad = 0;
    int Cin = 3;                         // input channels: 3 (RGB classifier) or 1 (ViTSTR, grey)
    int R = 1;                           // probability rows per image: 1 (cls token) or 25 (ViTSTR: tokens 0..24, vitstr.cpp:864-904)
    int tm = 128, tn = 128;
    const Tuning *tune = nullptr;        // per-device launch parameters (CU count, kernel selection), immutable
    int split_first = 0;                 // vitx_ctx_options::split_first: images of the first of two sub-batches (0 = the tile-round model)
#ifdef VITX_LAB
    int skip = 0;                        // VITX_SKIP (upper-bound experiments; results are garbage): 1 = no attention, 2 = no per-layer LayerNorm
#endif
    hipStream_t stream = nullptr;
    std::vector<void *> allocs;          // scratch of THIS context
    // weights: device copies are shared by every context of the same loaded model, device, operand type and quantisation mode
    // (WeightSet below; e.g. the two contexts of INTEGRATION.md's "two forwards in flight"): uploaded by the first, freed with the last
    struct WeightSet {
        int device = 0;
        std::vector<void *> allocs;
        float *cls = nullptr, *pos = nullptr, *pe_b = nullptr, *norm_w = nullptr, *norm_b = nullptr, *head_b = nullptr;
        void *pe_w = nullptr, *head_w = nullptr;
        QuantW head_q;
        std::vector<LayerW> layers;
        size_t weight_bytes = 0;
        ~WeightSet() {       // may run on any thread (the last context of the set): leave the caller's current device as it was
            int cur = -1; (void)hipGetDevice(&cur);
            (void)hipSetDevice(device); for (void *p : allocs) (void)hipFree(p);
            if (cur >= 0 && cur != device) (void)hipSetDevice(cur);
        }
    };
    std::shared_ptr<WeightSet> wset;
    bool weights_shared = false;         // this context found the set already uploaded (vitx_ctx_shares_weights)
    float *cls = nullptr, *pos = nullptr, *pe_b = nullptr, *norm_w = nullptr, *norm_b = nullptr, *head_b = nullptr;
    void *pe_w = nullptr, *head_w = nullptr;
    QuantW head_q;
    std::vector<LayerW> layers;
    // quantised files: vitx_ctx_options::quant_on_host restores the r01 behaviour (expand once on the host at upload, 16 bits per weight in HBM)
    bool quant_on_device = true;
    // q4_0 GEMMs with at most this many rows expand the blocks inside the GEMM's LDS-fill path (vitx_ctx_options::q4_fused_rows).  0 = never:
    // measured on ViT-B (profiles/r02c_quant.txt) the 128x128-tile fused kernel loses to "expand the layer just in time, then the skinny ring
    // kernels" at every batch size (batch 1: 1.75 vs 1.06 ms, batch 8: 2.15 vs 1.33 ms), so it is an option, not the default.
    int q4_fused_rows = 0;
    size_t weight_bytes = 0;             // device bytes held by weight matrices (vitx_ctx_weight_bytes)
    // LayerNorm fused into the residual GEMMs (GemmLn, kernels.h): norm2 rides in proj, the next layer's norm1 in fc2, wherever those GEMMs
    // run on the wide persistent kernel.  vitx_ctx_options::no_ln_fusion turns it off (every LayerNorm its own launch; same bits).
    // F16 = the parity mode: q, k, v stay f32-grade into the attention products, as the reference's do (vit.cpp:826-858).  The QKV GEMM then
    // emits two fp16 planes (EPI_BIAS_HILO) and the precise streaming kernel multiplies hi.hi + (hi.lo + lo.hi) / 2048 (attention_stream.hip).
    // Head dim 64 only (the generic head-dim kernel keeps fp16 q, k, v).
    bool prec_attn = false;
    // Last layer of a classifier: vit.cpp:910-911 reads row 0 of its output and nothing else, and rows meet each other only inside the attention
    // (through k and v).  So after the last qkv projection only the class token's row is carried on: its attention (attention_cls_kernel), then the
    // output projection, norm2 and the MLP on ONE row per image (Slice::Xc).  Same results; 0.76 of one layer's work is never asked for
    // (ViT-B: 6.3 % of the forward's flops).  vitx_ctx_options::last_layer_all_rows computes every row as the reference graph does (bench.py's headline does).
    // Not taken by ViTSTR contexts (25 rows per image feed the head) or while a residual-stream trace is on (the trace shows every row).
    bool cls_tail = true;
    bool ln_fuse = true;
    unsigned ln_epoch = 0;               // tag of the next fused launch (unique per launch; 0 is never used)
    unsigned ln_timeout = 20000;         // 200 us of the 100 MHz wall clock before a workgroup leaves its tile to the fix-up
    int ln_test = 0;                     // vitx_ctx_options::ln_test (parity tests: forced time-outs, GemmLn::test)
    int call_limit = 0;                  // images ONE pass of the kernels takes (32-bit byte offsets into the largest per-slice buffer); larger batches run as several passes
    // Fall-back budget (r03 advisor): a fused tile whose peers do not answer stalls up to ln_timeout per polled peer before it leaves its row block to
    // the consumer -- correct, but a throughput cliff when the peers' CUs are held by someone else (a second context, another process).  Every
    // forward copies the slices' fall-back counters to pinned host memory (asynchronously: the values read here are one forward old); more than
    // kLnBudget tiles per forward on average over a window of kLnWindow forwards switches the fusion off for this context (same bits either way).
    unsigned *ln_fb_host = nullptr;      // [nslices] pinned
    unsigned long long ln_fb_base = 0;   // counter total at the start of the current window
    int ln_fb_forwards = 0;
    bool ln_fuse_disabled = false;       // the budget tripped (vitx_ctx_ln_fusion_active)
    // ... and is re-armed after a cool-down (r04 advisor: one burst of contention -- another context warming up -- must not cost the fused path
    // for the rest of the context's life): the fusion is tried again after ln_cool_len forwards; every further trip doubles the cool-down (cap 2^16)
    int ln_cool_left = 0, ln_cool_len = 256;
    // activations: the batch is cut into `nslices` contiguous sub-batches, each with its own scratch and HIP stream,
    // so that the tail round / launch gaps / epilogues of one sub-batch's kernels are filled by the other's
    // (measured +10 % images/s at batch 256, tools/two_stream_probe.py).  Sub-batches are independent images.
    struct Slice {
        int cap = 0;                 // images this slice can hold
        float *X = nullptr;          // [Mpad][D] f32 residual stream
        void *U = nullptr;           // [Mpad][D] norm1 output / attention output
        void *U2 = nullptr;          // [Mpad][D] norm2 output (its own buffer: proj reads U while its epilogue writes the normalised rows)
        unsigned long long *ln_sync = nullptr;   // [Mpad / 256][D / 256][256][2] statistics granules of the fused LayerNorm
        unsigned *ln_todo = nullptr;             // [ln_blocks] row blocks left to the fix-up launch; [ln_blocks] = the fallback counter
        int ln_blocks = 0;                       // Mpad / 256 of the slice's capacity
        void *QKV = nullptr;         // [Mpad][3D]; the parity mode's lo plane follows at qkv_lo_off elements
        long qkv_lo_off = 0;
        void *Hbuf = nullptr;        // [Mpad][4D]  (also the im2col rows of the patch-embed GEMM)
        float *Xc = nullptr;         // [Bpad][D] f32 class-token rows of the residual stream through the last layer's tail (cls_tail)
        void *Z = nullptr;           // [Bpad][D] final-LN output of the cls rows
        void *Wq[W_PER_LAYER] = {nullptr, nullptr, nullptr, nullptr};   // just-in-time expansion of the current layer's quantised matrices
        void *Wq_head = nullptr;
        float *logits = nullptr;     // [Bpad][C_pad]
        hipStream_t stream = nullptr;
        hipEvent_t done = nullptr;
        Tuning tune;                 // the device's tuning with n_cu = the CUs this slice's stream may use (CU-masked streams)
    };
    int nslices = 1;
    std::vector<Slice> slices;
    hipEvent_t fork = nullptr;
    std::vector<hipStream_t> probed_streams;   // caller streams the internal streams were already checked against (ensure_concurrent): never re-probed
    int stream_retries = 0;               // internal streams re-created because they did not run beside the caller's stream
    hipEvent_t probe_a = nullptr, probe_b = nullptr;
    float *img = nullptr;        // [max_batch][S][S][3] staging for the host entry point
    float *probs = nullptr;      // [max_batch][C]
    float *logits_all = nullptr; // [max_batch][C] staging for the host entry point
    // residual-stream trace (vitx_trace_enable)
    std::vector<int> trace_ids;
    float *trace_buf = nullptr;  // [L + 1][n_ids][N][D]
    // hipGraph cache of the single-stream (small-batch) forward, opt-in (vitx_ctx_options::graph).  Key = (images, batch, outputs): the graph
    // bakes the pointers in.  An entry is captured the second time in a row its key is seen (one-off calls are never captured).
    // Measured (profiles/r02f/hipgraph_small_batch.txt): replaying the ~100 dependent launches as a graph takes the enqueue work off
    // the host thread but does not shorten the forward -- ViT-B batch 1: 0.874 vs 0.867 ms, batch 8: 1.141 vs 1.136 ms.  The chain is
    // bound by the GPU-side cost of ~100 dependent 5-12 us kernels, not by the host's launch rate, so it is not the default.
    struct GraphEntry { const void *imgs; void *probs, *logits; int n; hipGraphExec_t exec; };
    std::vector<GraphEntry> graphs;
    GraphEntry graph_last{nullptr, nullptr, nullptr, 0, nullptr};
    bool graphs_on = false;
    // profiling
    bool prof_on = false;
    hipEvent_t prof_base = nullptr;
    struct Rec { int cls; hipEvent_t a, b; double flops, bytes; };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;

    ~vitx_ctx() {
        (void)hipSetDevice(device);
        for (hipEvent_t e : ev_pool) (void)hipEventDestroy(e);
        for (auto &ge : graphs) (void)hipGraphExecDestroy(ge.exec);
        for (auto &sl : slices) { if (sl.stream) (void)hipStreamDestroy(sl.stream); if (sl.done) (void)hipEventDestroy(sl.done); }
        if (fork) (void)hipEventDestroy(fork);
        if (probe_a) (void)hipEventDestroy(probe_a);
        if (probe_b) (void)hipEventDestroy(probe_b);
        if (prof_base) (void)hipEventDestroy(prof_base);
        if (ln_fb_host) (void)hipHostFree(ln_fb_host);
        if (trace_buf) (void)hipFree(trace_buf);
        for (void *p : allocs) (void)hipFree(p);
        if (stream) (void)hipStreamDestroy(stream);
    }
    int dmalloc(void **p, size_t bytes, bool zero) {
        HIP_TRY(hipMalloc(p, bytes ? bytes : 16));
        allocs.push_back(*p);
        if (zero) HIP_TRY(hipMemset(*p, 0, bytes ? bytes : 16));
        return VITX_OK;
    }
    int wmalloc(void **p, size_t bytes) {      // weight storage: owned by the shared set
        HIP_TRY(hipMalloc(p, bytes ? bytes : 16));
        wset->allocs.push_back(*p);
        return VITX_OK;
    }
    hipEvent_t next_event() {
        if (ev_used == ev_pool.size()) { hipEvent_t e; (void)hipEventCreate(&e); ev_pool.push_back(e); }
        return ev_pool[ev_used++];
    }
};

namespace {

// f32 vector -> device f32 (padded with zeros to n_pad)
int upload_f32(vitx_ctx *c, const HostTensor *t, float **out, size_t n_pad = 0) {
    std::vector<float> h((size_t)t->nelements());
    t->decode_f32(h.data());
    if (n_pad > h.size()) h.resize(n_pad, 0.0f);
    int rc = c->wmalloc((void **)out, h.size() * 4);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(*out, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    return VITX_OK;
}

// [N][K] matrix -> operand type, rows padded to n_pad, cols to k_pad (zeros).  f16 file data is
// forwarded bit-exact in F16 mode; everything else is decoded to f32 and rounded once (RNE).
// patch_P > 0: the patch-embedding kernel [D][Cin * P * P]: its K axis is permuted to the image's memory order (patch_embed.hip)
int upload_matrix(vitx_ctx *c, const HostTensor *t, int Nrows, int K, int n_pad, int k_pad, void **out, int patch_P = 0, int patch_Cin = 0) {
    std::vector<uint16_t> h((size_t)n_pad * k_pad, 0);
    if (t->type == T_F16 && c->dtype == VITX_F16) {
        const uint16_t *src = (const uint16_t *)t->raw.data();
        for (int n = 0; n < Nrows; ++n) memcpy(&h[(size_t)n * k_pad], src + (size_t)n * K, (size_t)K * 2);
    } else {
        std::vector<float> f((size_t)Nrows * K);
        t->decode_f32(f.data());
        for (int n = 0; n < Nrows; ++n)
            for (int k = 0; k < K; ++k)
                h[(size_t)n * k_pad + k] = c->dtype == VITX_F16 ? f32_to_f16_bits(f[(size_t)n * K + k]) : f32_to_bf16_bits(f[(size_t)n * K + k]);
    }
    if (patch_P > 0) { std::vector<uint16_t> hp(h.size(), 0); patch_embed_permute_k(h.data(), hp.data(), Nrows, patch_Cin, patch_P, k_pad); h.swap(hp); }
    int rc = c->wmalloc(out, h.size() * 2);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(*out, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    c->weight_bytes += h.size() * 2;
    return VITX_OK;
}
int upload_quant(vitx_ctx *c, const HostTensor *t, int Nrows, int K, int n_pad, QuantW *q);
// A 2-D "*weight" tensor: block types stay quantised on the device (q), everything else is uploaded expanded (dense).
int upload_weight(vitx_ctx *c, const HostTensor *t, int Nrows, int K, int n_pad, void **dense, QuantW *q) {
    *dense = nullptr;
    if (c->quant_on_device) { int rc = upload_quant(c, t, Nrows, K, n_pad, q); if (rc) return rc; }
    if (q->blocks) return VITX_OK;
    return upload_matrix(c, t, Nrows, K, n_pad, K, dense);
}

// Quantised [N][K] matrix -> device, still in block form.  Returns VITX_OK with q->blocks == nullptr when the tensor is not a
// block type (the caller then uploads the expanded matrix).
int upload_quant(vitx_ctx *c, const HostTensor *t, int Nrows, int K, int n_pad, QuantW *q) {
    const int bb = type_block_bytes(t->type);
    if (t->type == T_F32 || t->type == T_F16 || !bb || K % 32) return VITX_OK;
    const size_t nbk = (size_t)K / 32;
    q->type = t->type; q->N = Nrows; q->K = K; q->n_pad = n_pad;
    int rc;
    if (t->type == T_Q4_0) {        // split planes, rows padded (zero scales -> the pad rows expand to zeros)
        std::vector<uint8_t> qs((size_t)n_pad * nbk * 16, 0);
        std::vector<uint16_t> ds((size_t)n_pad * nbk, 0);
        const uint8_t *src = t->raw.data();
        for (size_t b = 0; b < (size_t)Nrows * nbk; ++b) { memcpy(&ds[b], src + b * 18, 2); memcpy(&qs[b * 16], src + b * 18 + 2, 16); }
        if ((rc = c->wmalloc(&q->blocks, qs.size()))) return rc;
        if ((rc = c->wmalloc((void **)&q->scales, ds.size() * 2))) return rc;
        HIP_TRY(hipMemcpy(q->blocks, qs.data(), qs.size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(q->scales, ds.data(), ds.size() * 2, hipMemcpyHostToDevice));
        c->weight_bytes += qs.size() + ds.size() * 2;
    } else {
        const size_t bytes = (size_t)Nrows * nbk * bb;
        if ((rc = c->wmalloc(&q->blocks, bytes))) return rc;
        HIP_TRY(hipMemcpy(q->blocks, t->raw.data(), bytes, hipMemcpyHostToDevice));
        c->weight_bytes += bytes;
    }
    return VITX_OK;
}

struct ProfScope {
    vitx_ctx *c; hipStream_t s; size_t idx = 0; bool on;
    ProfScope(vitx_ctx *c_, hipStream_t s_, int cls, double flops, double bytes) : c(c_), s(s_), on(c_->prof_on) {
        if (!on) return;
        vitx_ctx::Rec r{cls, c->next_event(), c->next_event(), flops, bytes};
        idx = c->recs.size(); c->recs.push_back(r);
        (void)hipEventRecord(r.a, s);
    }
    ~ProfScope() { if (on) (void)hipEventRecord(c->recs[idx].b, s); }
};

// `fused` != nullptr: W is that q4_0 matrix and the GEMM expands the blocks in its own LDS-fill path (small batches).
// `fix`: the GemmLn of the LayerNorm-fusing GEMM that produced A (GemmArgs::fix); when the kernel this shape selects cannot recompute the
// row blocks that GEMM left behind, they are fixed by a launch of their own first.
int gemm(vitx_ctx *c, const Tuning &tune, hipStream_t st, int pc, int epi, const void *A, const void *W, const float *bias, void *out, const float *pos,
         int M, int M_real, int N, int N_pad, int K, int lda, int ldw, int ldo, int tpi, size_t out_elem_bytes, const QuantW *fused = nullptr, const GemmLn *ln = nullptr,
         const GemmLn *fix = nullptr, long hilo_off = 0, int rows_alg = 0) {
    GemmArgs a{};
    a.A = A; a.W = W; a.bias = bias; a.out = out; a.pos = pos; a.hilo_off = hilo_off;
    a.M = M; a.M_real = M_real; a.N = N; a.N_pad = N_pad; a.K = K; a.lda = lda; a.ldw = ldw; a.ldo = ldo; a.tpi = tpi;
    a.ln = ln;
    if (fix) {
        if (!fused && gemm_fix_capable(tune, a)) a.fix = fix;
        else {
            ProfScope ps(c, st, PC_LAYERNORM, 0, 0);
            HIP_TRY(launch_layernorm_fixup(c->dtype, fix->x, fix->w, fix->b, fix->out, M, K, fix->eps, fix->todo, fix->epoch, st));
        }
    }
    // algorithmic work of the launch: the REAL rows (a LayerNorm-fusing launch computes and stores its pad rows too -- GemmLn -- but they are not work
    // the forward asked for: r03 counted them, +0.46 % on the fc2 figure)
    const int M_alg = rows_alg > 0 ? rows_alg : M_real;
    double bytes = (double)M_alg * K * 2 + (double)N * K * (fused ? 0.5625 : 2.0) + (double)M_alg * N * out_elem_bytes;
    if (epi == EPI_BIAS_RESID) bytes += (double)M_alg * N * 4;
    if (epi == EPI_BIAS_HILO) bytes += (double)M_alg * N * out_elem_bytes;        // the second plane
    if (ln) bytes += (double)M_alg * N * 2;
    ProfScope ps(c, st, pc, 2.0 * M_alg * (double)N * K, bytes);
    if (fused) {
        a.W = fused->blocks; a.Wscale = fused->scales;
        HIP_TRY(launch_gemm_q4(c->dtype, epi, a, st));
        return VITX_OK;
    }
    HIP_TRY(launch_gemm(tune, c->dtype, epi, a, st));
    return VITX_OK;
}

}  // namespace

extern "C" {

int vitx_ctx_create(const vitx_model *m, int device, int max_batch, int dtype, vitx_ctx **out) { return vitx_ctx_create_ex(m, device, max_batch, dtype, nullptr, out); }

int vitx_ctx_create_ex(const vitx_model *m, int device, int max_batch, int dtype, const vitx_ctx_options *opt_in, vitx_ctx **out) {
    if (!m || !out || max_batch <= 0 || (dtype != VITX_F16 && dtype != VITX_BF16)) { set_error("vitx_ctx_create: invalid argument"); return VITX_ERR_ARG; }
    *out = nullptr;
    vitx_ctx_options opt{};                   // all zero = every default
    if (opt_in) {
        if (opt_in->struct_size < 8 || opt_in->struct_size > (int)sizeof(vitx_ctx_options)) { set_error("vitx_ctx_create_ex: options.struct_size %d is not a size this library knows", opt_in->struct_size); return VITX_ERR_ARG; }
        memcpy(&opt, opt_in, (size_t)opt_in->struct_size);
        if (opt.streams < 0 || opt.streams > 4 || opt.q4_fused_rows < 0 || opt.split_first < 0 || (opt.last_layer_all_rows & ~1)) { set_error("vitx_ctx_create_ex: option out of range"); return VITX_ERR_ARG; }
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("vitx_ctx_create: no HIP device available (this engine has no CPU fallback)"); return VITX_ERR_HIP; }
    if (device < 0 || device >= ndev) { set_error("vitx_ctx_create: device %d out of range (%d devices)", device, ndev); return VITX_ERR_ARG; }
    const vitx_hparams &hp = m->hp;
    if (hp.num_attention_heads <= 0 || hp.hidden_size % hp.num_attention_heads) { set_error("vitx_ctx_create: hidden_size %d is not a multiple of %d heads", hp.hidden_size, hp.num_attention_heads); return VITX_ERR_UNSUPPORTED; }
    if (hp.hidden_size % 64) { set_error("vitx_ctx_create: hidden_size must be a multiple of 64"); return VITX_ERR_UNSUPPORTED; }
    HIP_TRY(hipSetDevice(device));
    std::unique_ptr<vitx_ctx> c(new (std::nothrow) vitx_ctx());
    if (!c) return VITX_ERR_NOMEM;
    c->model = m; c->hp = hp; c->device = device; c->dtype = dtype; c->max_batch = max_batch;
    c->D = hp.hidden_size; c->L = hp.num_hidden_layers; c->H = hp.num_attention_heads; c->C = hp.num_classes; c->P = hp.patch_size; c->S = hp.img_size;
    c->Cin = m->in_chans; c->R = m->in_chans == 1 ? VITX_VITSTR_SEQ_LEN : 1;
    c->g = c->S / c->P; c->N = c->g * c->g + 1; c->Kpe = c->Cin * c->P * c->P; c->Kpe_pad = round_up(c->Kpe, 64);
    if (c->N < c->R) { set_error("vitx_ctx_create: a ViTSTR head reads %d tokens, this model has %d (img_size %d, patch_size %d)", c->R, c->N, c->S, c->P); return VITX_ERR_UNSUPPORTED; }
    c->tm = gemm_tile_m(); c->tn = gemm_tile_n();
    c->C_pad = round_up(c->C, c->tn);
    // validate against what the kernels are actually instantiated for (a context that would fail on its first forward is refused here)
    if (!attention_supports(c->N, c->D, c->H)) {
        set_error("vitx_ctx_create: attention needs a head_dim that is a multiple of 8 up to 128 (this model: %d) and at least one token (%d tokens, img_size %d, patch_size %d)", c->D / c->H, c->N, c->S, c->P);
        return VITX_ERR_UNSUPPORTED;
    }
    if (!layernorm_supports(c->D)) { set_error("vitx_ctx_create: hidden_size %d has no LayerNorm instantiation (64, 128, 192, 256, 320, 384, 448, 512, 576, 640, 768, 896, 1024, 1152, 1280, 1408, 1536, 1664, 2048)", c->D); return VITX_ERR_UNSUPPORTED; }
    c->tune = tuning_for_device(device);
    if (!c->tune) { set_error("vitx_ctx_create: kernel bring-up on device %d failed: %s", device, hipGetErrorString(hipGetLastError())); return VITX_ERR_HIP; }
    c->split_first = opt.split_first;
    c->prec_attn = dtype == VITX_F16 && c->D == c->H * 64 && !opt.f16_fast_attention;
    c->quant_on_device = !opt.quant_on_host;
    c->cls_tail = !opt.last_layer_all_rows && c->R == 1 && attention_cls_supports(c->N, c->D, c->H);
    c->q4_fused_rows = opt.q4_fused_rows;
    c->graphs_on = opt.graph != 0;
    // fault injection for the parity tests: honoured only with the key in the upper half (VITX_LN_TEST_KEY | mode), so that no caller sets it by accident
    if (opt.ln_test) {
        if ((opt.ln_test & (int32_t)0xffff0000) != (int32_t)VITX_LN_TEST_KEY) { set_error("vitx_ctx_create_ex: ln_test is a test-only switch (it needs its key: include/vitx.h)"); return VITX_ERR_ARG; }
        c->ln_test = opt.ln_test & 0xffff;
        if (c->ln_test & ~7) { set_error("vitx_ctx_create_ex: ln_test mode %d has bits outside 1 | 2 | 4", c->ln_test); return VITX_ERR_ARG; }
        if ((c->ln_test & 3) == 3) c->ln_timeout = 5000;       // real time-outs in the test: 50 us (with or without bit 4; the kernel tests the bits one by one too)
    }
    c->ln_fuse = !opt.no_ln_fusion && !opt.graph;        // a captured launch would replay its epoch tag: no fusion under the graph cache
#ifdef VITX_LAB
    if (const char *e = getenv("VITX_SKIP")) c->skip = atoi(e);
    if (const char *e = getenv("VITX_SPLIT")) c->split_first = atoi(e);
#endif
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));

    const int D = c->D, tn = c->tn;
    int rc;
    auto T = [&](const std::string &n) { return m->find(n); };
    // device copies of the weights: one set per (loaded model, device, operand type, block mode), shared by every context that asks for it
    static std::mutex wreg_mu;
    static std::map<std::tuple<uint64_t, int, int, int>, std::weak_ptr<vitx_ctx::WeightSet>> wreg;
    const auto wkey = std::make_tuple(m->uid, device, dtype, c->quant_on_device ? 1 : 0);
    std::unique_lock<std::mutex> wlock(wreg_mu);           // held through the upload: a second context of the same model waits for the first
    for (auto it = wreg.begin(); it != wreg.end();) it = it->second.expired() ? wreg.erase(it) : std::next(it);      // sets whose last context is gone
    if (auto have = wreg[wkey].lock()) {
        c->wset = have; c->weights_shared = true;
    } else {
        c->wset = std::make_shared<vitx_ctx::WeightSet>();
        c->wset->device = device;
    }
    if (!c->weights_shared) {
    if ((rc = upload_f32(c.get(), T("cls_token"), &c->cls))) return rc;
    if ((rc = upload_f32(c.get(), T("pos_embed"), &c->pos))) return rc;
    if ((rc = upload_f32(c.get(), T("patch_embed.proj.bias"), &c->pe_b, round_up(D, tn)))) return rc;
    if ((rc = upload_matrix(c.get(), T("patch_embed.proj.weight"), D, c->Kpe, round_up(D, tn), c->Kpe_pad, &c->pe_w, c->P, c->Cin))) return rc;
    c->layers.resize(c->L);
    for (int i = 0; i < c->L; ++i) {
        const std::string p = "blocks." + std::to_string(i) + ".";
        LayerW &w = c->layers[i];
        if ((rc = upload_f32(c.get(), T(p + "norm1.weight"), &w.ln1_w))) return rc;
        if ((rc = upload_f32(c.get(), T(p + "norm1.bias"), &w.ln1_b))) return rc;
        if ((rc = upload_f32(c.get(), T(p + "norm2.weight"), &w.ln2_w))) return rc;
        if ((rc = upload_f32(c.get(), T(p + "norm2.bias"), &w.ln2_b))) return rc;
        if ((rc = upload_f32(c.get(), T(p + "attn.qkv.bias"), &w.qkv_b, round_up(3 * D, tn)))) return rc;
        if ((rc = upload_f32(c.get(), T(p + "attn.proj.bias"), &w.proj_b, round_up(D, tn)))) return rc;
        if ((rc = upload_f32(c.get(), T(p + "mlp.fc1.bias"), &w.fc1_b, round_up(4 * D, tn)))) return rc;
        if ((rc = upload_f32(c.get(), T(p + "mlp.fc2.bias"), &w.fc2_b, round_up(D, tn)))) return rc;
        if ((rc = upload_weight(c.get(), T(p + "attn.qkv.weight"), 3 * D, D, round_up(3 * D, tn), &w.qkv_w, &w.q[W_QKV]))) return rc;
        if ((rc = upload_weight(c.get(), T(p + "attn.proj.weight"), D, D, round_up(D, tn), &w.proj_w, &w.q[W_PROJ]))) return rc;
        if ((rc = upload_weight(c.get(), T(p + "mlp.fc1.weight"), 4 * D, D, round_up(4 * D, tn), &w.fc1_w, &w.q[W_FC1]))) return rc;
        if ((rc = upload_weight(c.get(), T(p + "mlp.fc2.weight"), D, 4 * D, round_up(D, tn), &w.fc2_w, &w.q[W_FC2]))) return rc;
    }
    if ((rc = upload_f32(c.get(), T("norm.weight"), &c->norm_w))) return rc;
    if ((rc = upload_f32(c.get(), T("norm.bias"), &c->norm_b))) return rc;
    if ((rc = upload_f32(c.get(), T("head.bias"), &c->head_b, c->C_pad))) return rc;
    if ((rc = upload_weight(c.get(), T("head.weight"), c->C, D, c->C_pad, &c->head_w, &c->head_q))) return rc;
        vitx_ctx::WeightSet &w = *c->wset;
        w.cls = c->cls; w.pos = c->pos; w.pe_b = c->pe_b; w.norm_w = c->norm_w; w.norm_b = c->norm_b; w.head_b = c->head_b;
        w.pe_w = c->pe_w; w.head_w = c->head_w; w.head_q = c->head_q; w.layers = c->layers; w.weight_bytes = c->weight_bytes;
        wreg[wkey] = c->wset;
    } else {
        const vitx_ctx::WeightSet &w = *c->wset;
        c->cls = w.cls; c->pos = w.pos; c->pe_b = w.pe_b; c->norm_w = w.norm_w; c->norm_b = w.norm_b; c->head_b = w.head_b;
        c->pe_w = w.pe_w; c->head_w = w.head_w; c->head_q = w.head_q; c->layers = w.layers; c->weight_bytes = w.weight_bytes;
    }
    wlock.unlock();

    // sub-batch slices (vitx_ctx_options::streams; 1 = single stream).  Small contexts stay single-slice.
    int ns = opt.streams > 0 ? opt.streams : 2;
    if (ns < 1) ns = 1;
    if (ns > 4) ns = 4;
    if (max_batch < 8 * ns) ns = 1;
    c->nslices = ns;
    c->slices.resize(ns);
    const size_t hcols = std::max<size_t>((size_t)4 * D, (size_t)c->Kpe_pad);
    {
        // The kernels address every activation buffer with 32-bit BYTE offsets (buffer instructions): a sub-batch must keep its largest buffer --
        // the MLP hidden tensor, or the two QKV planes of the F16 parity mode -- below 0xf0000000 bytes (ViT-B: 3326 images per sub-batch, 2217 in
        // parity mode).  Batches beyond one such window run as several passes through the same scratch (vitx_forward_device), so max_batch
        // itself is only bounded by memory.  r04: the guards used to sit in the individual launchers only, and a 10 000-image batch computed garbage.
        const size_t row_bytes = std::max<size_t>(hcols * 2, (size_t)3 * D * 2 * (c->prec_attn ? 2 : 1));
        const size_t rows = (size_t)0xf0000000u / row_bytes / 256 * 256;
        const long per_slice = (long)(rows / c->N);
        if (per_slice < 1) { set_error("vitx_ctx_create: a single image exceeds the kernels' 32-bit buffer window (%d tokens x %d)", c->N, D); return VITX_ERR_UNSUPPORTED; }
        c->call_limit = (int)std::min<long>((long)max_batch, per_slice);          // whatever the split of a pass, no sub-batch exceeds the window
    }
    for (int i = 0; i < ns; ++i) {
        vitx_ctx::Slice &sl = c->slices[i];
        sl.cap = std::min(max_batch, c->call_limit);     // every slice can hold a whole pass: the split point is chosen per call (split_batch)
        const size_t Mpad = (size_t)round_up(sl.cap * c->N, c->tm), Bpad = (size_t)round_up(sl.cap * c->R, c->tm);
        if ((rc = c->dmalloc((void **)&sl.X, Mpad * D * 4, true))) return rc;
        if ((rc = c->dmalloc(&sl.U, Mpad * D * 2, true))) return rc;
        if ((rc = c->dmalloc(&sl.U2, Mpad * D * 2, true))) return rc;
        if (D % 256 == 0 && D / 256 <= 4) {
            if ((rc = c->dmalloc((void **)&sl.ln_sync, (Mpad / 256) * (size_t)(D / 256) * 256 * 2 * sizeof(unsigned long long), true))) return rc;
            if ((rc = c->dmalloc((void **)&sl.ln_todo, (Mpad / 256 + 1) * sizeof(unsigned), true))) return rc;
            sl.ln_blocks = (int)(Mpad / 256);
        }
        if ((rc = c->dmalloc(&sl.QKV, Mpad * 3 * D * 2 * (c->prec_attn ? 2 : 1), true))) return rc;
        sl.qkv_lo_off = c->prec_attn ? (long)(Mpad * 3 * D) : 0;       // capacity; a forward places the lo plane right behind ITS rows (forward_slice)
        if ((rc = c->dmalloc(&sl.Hbuf, Mpad * hcols * 2, true))) return rc;
        if ((rc = c->dmalloc(&sl.Z, Bpad * D * 2, true))) return rc;
        if (c->cls_tail && (rc = c->dmalloc((void **)&sl.Xc, Bpad * D * 4, true))) return rc;
        if ((rc = c->dmalloc((void **)&sl.logits, Bpad * c->C_pad * 4, true))) return rc;
        // expansion scratch for quantised matrices: one buffer per matrix kind, shared by all layers (the largest layer decides)
        for (int k = 0; k < W_PER_LAYER; ++k) {
            size_t need = 0;
            for (const LayerW &w : c->layers) if (w.q[k].blocks) need = std::max(need, (size_t)w.q[k].n_pad * w.q[k].K * 2);
            if (need && (rc = c->dmalloc(&sl.Wq[k], need, false))) return rc;
        }
        if (c->head_q.blocks && (rc = c->dmalloc(&sl.Wq_head, (size_t)c->head_q.n_pad * c->head_q.K * 2, false))) return rc;
        sl.tune = *c->tune;
#ifdef VITX_LAB
        if (const char *e = getenv("VITX_SLICE_CU")) { if (ns > 1 && atoi(e) > 0) sl.tune.n_cu = atoi(e); }      // persistent grids of each sub-batch capped (experiment)
#endif
        if (ns > 1 && i > 0) {
            // Slice 0 runs on the CALLER's stream, slices 1.. on internal HIGH-priority streams.  The runtime multiplexes all streams of one
            // priority onto a small pool of hardware queues (GPU_MAX_HW_QUEUES, 4 by default), round-robin in creation order; a hardware queue
            // executes its packets in order.  r02 gave slice 0 its own normal-priority stream: whenever that stream shared a hardware queue
            // with the caller's (torch's pool of streams, a second context in the process ...), step k + 1's slice-0 kernels queued up behind
            // the caller stream's wait for step k's slice 1, and the two sub-batches ran back to back -- measured r03: the 2nd and the 6th
            // context created in one process ran 11.8 instead of 9.9 ms per forward.  Pools are per priority: the caller's stream (normal,
            // unless the caller chose otherwise) and the internal ones (high, created back to back) can never share a queue.
            int least = 0, greatest = 0;
            HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
            HIP_TRY(hipStreamCreateWithPriority(&sl.stream, hipStreamNonBlocking, greatest));
            HIP_TRY(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
        }
    }
    if (ns > 1) HIP_TRY(hipEventCreateWithFlags(&c->fork, hipEventDisableTiming));
    HIP_TRY(hipHostMalloc((void **)&c->ln_fb_host, sizeof(unsigned) * 4, hipHostMallocDefault));
    for (int i = 0; i < 4; ++i) c->ln_fb_host[i] = 0;
    if ((rc = c->dmalloc((void **)&c->img, (size_t)max_batch * c->S * c->S * c->Cin * 4, false))) return rc;
    if ((rc = c->dmalloc((void **)&c->probs, (size_t)max_batch * c->R * c->C * 4, true))) return rc;
    if ((rc = c->dmalloc((void **)&c->logits_all, (size_t)max_batch * c->R * c->C * 4, true))) return rc;
    HIP_TRY(hipDeviceSynchronize());
    *out = c.release();
    return VITX_OK;
}

void vitx_ctx_free(vitx_ctx *c) { delete c; }
int vitx_ctx_max_batch(const vitx_ctx *c) { return c ? c->max_batch : 0; }
static void split_batch(const vitx_ctx *c, int n, int ns, int *m);
int vitx_ctx_split(const vitx_ctx *c, int n, int32_t *images, int max_parts) {
    if (!c || !images || max_parts <= 0 || n <= 0 || n > c->max_batch) return 0;
    n = std::min(n, c->call_limit);              // a batch beyond the kernels' window runs as several passes: this is the first one's cut
    const int ns = (c->nslices > 1 && n >= 8 * c->nslices) ? c->nslices : 1;
    if (ns > max_parts) return 0;
    int m[4] = {n, 0, 0, 0};
    if (ns > 1) split_batch(c, n, ns, m);
    for (int i = 0; i < ns; ++i) images[i] = m[i];
    return ns;
}
int vitx_ctx_out_rows(const vitx_ctx *c) { return c ? c->R : 0; }

static int forward_slice(vitx_ctx *c, vitx_ctx::Slice &sl, hipStream_t st, const void *d_imgs, int first_img, int n, void *d_probs, void *d_logits) {
    // residual-stream trace: copy X of the traced images that live in this sub-batch (stage 0 = after patch embedding, il + 1 = after layer il)
    auto trace = [&](int stage) -> int {
        const size_t per = (size_t)c->N * c->D;
        for (size_t k = 0; k < c->trace_ids.size(); ++k) {
            const int id = c->trace_ids[k];
            if (id < first_img || id >= first_img + n) continue;
            HIP_TRY(hipMemcpyAsync(c->trace_buf + ((size_t)stage * c->trace_ids.size() + k) * per, sl.X + (size_t)(id - first_img) * per, per * 4, hipMemcpyDeviceToDevice, st));
        }
        return VITX_OK;
    };
    const int D = c->D, N = c->N, tm = c->tm, tn = c->tn, dt = c->dtype;
    const int tpi = c->g * c->g;
    const Tuning &tn_ = sl.tune;
    const int Mp_real = n * tpi;                                   // patch rows
    const int M_real = n * N, M = round_up(M_real, tm);            // token rows
    const double eb = 2.0;                                          // operand bytes
    const long lo_off = c->prec_attn ? (long)M * 3 * D : 0;         // F16 parity mode: the lo plane of q, k, v right behind this sub-batch's hi plane (elements)

    // patch embedding (vit.cpp:747-797) in one launch: im2col gather, GEMM, + bias + pos, token scatter, class rows (patch_embed.hip)
    int rc;
    {
        ProfScope ps(c, st, PC_GEMM_PATCH, 2.0 * Mp_real * (double)D * c->Kpe, (double)n * c->S * c->S * c->Cin * 4 + (double)M_real * D * 4);
        HIP_TRY(launch_patch_embed(dt, (const float *)d_imgs, c->pe_w, c->pe_b, c->pos, c->cls, sl.X, n, c->S, c->P, c->Cin, D, round_up(D, tn), c->Kpe_pad, st));
    }
    if (!c->trace_ids.empty() && (rc = trace(0))) return rc;
    // Quantised matrices (block form in HBM): a q4_0 GEMM with few rows expands the blocks in its own LDS-fill path; everything else
    // is expanded just in time, one launch per layer, into the slice's scratch and then streamed by the wide-tile kernels.
    auto fused_ok = [&](const QuantW &q, int rows) { return q.blocks && q.type == T_Q4_0 && rows <= c->q4_fused_rows && rows % 128 == 0 && q.n_pad % 128 == 0 && q.K % 64 == 0; };
    auto expand = [&](const QuantW *const *qs, void *const *dst, int count) -> int {
        bool done[W_PER_LAYER] = {false, false, false, false};
        for (int k = 0; k < count; ++k) {
            if (done[k] || !qs[k]) continue;
            DequantJob jobs[4]; int nj = 0; double bytes = 0;
            for (int m = k; m < count; ++m) {
                if (done[m] || !qs[m] || qs[m]->type != qs[k]->type) continue;
                jobs[nj++] = DequantJob{qs[m]->blocks, qs[m]->scales, dst[m], qs[m]->N, qs[m]->n_pad, qs[m]->K / 32};
                bytes += (double)qs[m]->N * qs[m]->K / 32 * type_block_bytes(qs[m]->type) + (double)qs[m]->n_pad * qs[m]->K * eb;
                done[m] = true;
            }
            ProfScope ps(c, st, PC_DEQUANT, 0, bytes);
            HIP_TRY(launch_dequant(dt, qs[k]->type, jobs, nj, st));
        }
        return VITX_OK;
    };
#ifdef VITX_LAB
    const int skip = c->skip;
#else
    constexpr int skip = 0;
#endif
    // LayerNorm fusion: decided per forward (the GEMM shape of this sub-batch must take the wide persistent kernel; never while the caller is
    // capturing a graph -- the epoch tag of a captured launch would be replayed).  The padded rows M_real .. M of X are then computed and
    // stored as well (GemmLn): they belong to this slice's scratch, start as zeros and stay finite.
    bool fuse = false;
    if (c->ln_fuse && sl.ln_sync) {
        GemmArgs probe{}; probe.M = M; probe.N = D; probe.N_pad = round_up(D, tn); probe.K = D; probe.lda = D; probe.ldw = D; probe.ldo = D;
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusActive; }
        GemmArgs probe2 = probe; probe2.K = 4 * D; probe2.lda = 4 * D; probe2.ldw = 4 * D;
        fuse = cs == hipStreamCaptureStatusNone && gemm_ln_fusable(tn_, probe) && gemm_ln_fusable(tn_, probe2);
    }
    // residual GEMM (+ the LayerNorm that follows it, fused when `fuse`; otherwise its own launch) -- proj + norm2, fc2 + the next norm1
    // `pend` receives the launch's GemmLn when the LayerNorm was fused: the GEMM that consumes ln_out next gets it as its `fix` argument
    auto resid_gemm_ln = [&](int pc, const void *A, const void *W, const float *bias, int K, const QuantW *fq, const float *lw, const float *lb, void *ln_out, GemmLn *pend) -> int {
        int rc2;
        pend->todo = nullptr;
        if (fuse && lw && !fq) {
            GemmLn ln{};
            ln.w = lw; ln.b = lb; ln.x = sl.X; ln.out = ln_out; ln.eps = c->hp.eps; ln.sync = sl.ln_sync; ln.todo = sl.ln_todo; ln.fallbacks = sl.ln_todo + sl.ln_blocks;
            if (++c->ln_epoch == 0) c->ln_epoch = 1;
            ln.epoch = c->ln_epoch; ln.timeout = c->ln_timeout; ln.test = c->ln_test;
            if ((rc2 = gemm(c, tn_, st, pc, EPI_BIAS_RESID, A, W, bias, sl.X, nullptr, M, M, D, round_up(D, tn), K, K, K, D, 0, 4, nullptr, &ln, nullptr, 0, M_real))) return rc2;
            *pend = ln;
            return VITX_OK;
        }
        if ((rc2 = gemm(c, tn_, st, pc, EPI_BIAS_RESID, A, W, bias, sl.X, nullptr, M, M_real, D, round_up(D, tn), K, K, K, D, 0, 4, fq))) return rc2;
        if (lw) {
            ProfScope ps(c, st, PC_LAYERNORM, 0, (double)M_real * D * (4 + eb));
            if (!(skip & 2)) HIP_TRY(launch_layernorm(dt, sl.X, D, lw, lb, ln_out, D, M_real, D, c->hp.eps, st));
        }
        return VITX_OK;
    };
    GemmLn fix_u{}, fix_u2{};          // fused LayerNorm launches whose output (U / U2) has not been consumed yet
    const bool tail = c->cls_tail && c->trace_ids.empty();      // the last layer carries only the class-token rows past its qkv projection (vitx_ctx::cls_tail)
    const int Mc = round_up(n, tm);                             // rows of the tail GEMMs (n real ones)
    for (int il = 0; il < c->L; ++il) {
        const LayerW &w = c->layers[il];
        const bool tail_now = tail && il + 1 == c->L;
        const void *Wl[W_PER_LAYER] = {w.qkv_w, w.proj_w, w.fc1_w, w.fc2_w};
        const QuantW *Fl[W_PER_LAYER] = {nullptr, nullptr, nullptr, nullptr};      // matrices the fused kernel takes
        {
            const QuantW *todo[W_PER_LAYER] = {nullptr, nullptr, nullptr, nullptr};
            bool any = false;
            for (int k = 0; k < W_PER_LAYER; ++k) {
                if (!w.q[k].blocks) continue;
                if (fused_ok(w.q[k], (tail_now && k != W_QKV) ? Mc : M)) Fl[k] = &w.q[k];
                else { todo[k] = &w.q[k]; Wl[k] = sl.Wq[k]; any = true; }
            }
            if (any && (rc = expand(todo, sl.Wq, W_PER_LAYER))) return rc;
        }
        if (il == 0) {   // norm1 of the first layer (vit.cpp:808-812); every later norm1 comes out of the previous layer's fc2
            ProfScope ps(c, st, PC_LAYERNORM, 0, (double)M_real * D * (4 + eb));
            if (!(skip & 2)) HIP_TRY(launch_layernorm(dt, sl.X, D, w.ln1_w, w.ln1_b, sl.U, D, M_real, D, c->hp.eps, st));
        }
#ifdef VITX_LAB
        if (il == 0 && getenv("VITX_SNAP")) {      // lab: X and the first norm1 output of this slice
            static int run1[4] = {0, 0, 0, 0};
            const int si = (int)(&sl - &c->slices[0]);
            std::vector<char> hx((size_t)M_real * D * 4), hu((size_t)M_real * D * 2);
            HIP_TRY(hipMemcpyAsync(hx.data(), sl.X, hx.size(), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(hu.data(), sl.U, hu.size(), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            char fn[512];
            snprintf(fn, sizeof fn, "%s/snap_s%d_r%d_x0.bin", getenv("VITX_SNAP"), si, run1[si]); { FILE *f = fopen(fn, "wb"); if (f) { fwrite(hx.data(), 1, hx.size(), f); fclose(f); } }
            snprintf(fn, sizeof fn, "%s/snap_s%d_r%d_ln1.bin", getenv("VITX_SNAP"), si, run1[si]); { FILE *f = fopen(fn, "wb"); if (f) { fwrite(hu.data(), 1, hu.size(), f); fclose(f); } }
            ++run1[si];
        }
#endif
        // qkv projection (vit.cpp:820-821); `fix_u`: row blocks of U the previous layer's fc2 left to the fix-up are normalised in its prologue
#ifdef VITX_LAB
        static const int prec_dbg = getenv("VITX_PREC_DBG") ? atoi(getenv("VITX_PREC_DBG")) : 0;      // 1: HILO GEMM + fast attention on the hi plane; 2: plain GEMM + precise attention (lo plane stays zero)
#else
        constexpr int prec_dbg = 0;
#endif
        if ((rc = gemm(c, tn_, st, PC_GEMM_QKV, (c->prec_attn && prec_dbg != 2) ? EPI_BIAS_HILO : EPI_BIAS, sl.U, Wl[W_QKV], w.qkv_b, sl.QKV, nullptr, M, M_real, 3 * D, round_up(3 * D, tn), D, D, D, 3 * D, 0, 2, Fl[W_QKV], nullptr,
                       fix_u.todo ? &fix_u : nullptr, lo_off))) return rc;
        if (tail_now) {
            {   // attention of token 0 (vit.cpp:848-858 for the one row vit.cpp:910-911 keeps) -> compact rows U[b]; class rows of X -> Xc[b]
                ProfScope ps(c, st, PC_ATTENTION_CLS, 4.0 * n * c->H * (double)N * (D / c->H), (double)M_real * 2 * D * eb * (c->prec_attn ? 2 : 1) + (double)n * D * (eb + 8));
                HIP_TRY(launch_attention_cls(dt, sl.QKV, lo_off, sl.U, sl.X, sl.Xc, n, N, D, c->H, st));
            }
            // output projection + residual, norm2, MLP (vit.cpp:868-900) on the n class rows
            if ((rc = gemm(c, tn_, st, PC_GEMM_TAIL, EPI_BIAS_RESID, sl.U, Wl[W_PROJ], w.proj_b, sl.Xc, nullptr, Mc, n, D, round_up(D, tn), D, D, D, D, 0, 4, Fl[W_PROJ]))) return rc;
            {
                ProfScope ps(c, st, PC_LAYERNORM, 0, (double)n * D * (4 + eb));
                HIP_TRY(launch_layernorm(dt, sl.Xc, D, w.ln2_w, w.ln2_b, sl.U2, D, n, D, c->hp.eps, st));
            }
            if ((rc = gemm(c, tn_, st, PC_GEMM_TAIL, EPI_BIAS_GELU, sl.U2, Wl[W_FC1], w.fc1_b, sl.Hbuf, nullptr, Mc, n, 4 * D, round_up(4 * D, tn), D, D, D, 4 * D, 0, 2, Fl[W_FC1]))) return rc;
            if ((rc = gemm(c, tn_, st, PC_GEMM_TAIL, EPI_BIAS_RESID, sl.Hbuf, Wl[W_FC2], w.fc2_b, sl.Xc, nullptr, Mc, n, D, round_up(D, tn), 4 * D, 4 * D, 4 * D, D, 0, 4, Fl[W_FC2]))) return rc;
            break;
        }
        {   // attention (vit.cpp:826-866)
            ProfScope ps(c, st, PC_ATTENTION, 4.0 * n * c->H * (double)N * N * (D / c->H), (double)M_real * (c->prec_attn ? 7 : 4) * D * eb);
            if (!(skip & 1)) {
                if (c->prec_attn && prec_dbg != 1) {
                    if (prec_dbg == 3) HIP_TRY(hipDeviceSynchronize());          // lab: the attention kernel runs alone on the device
                    HIP_TRY(launch_attention_stream(dt, true, sl.QKV, sl.U, n, N, D, c->H, lo_off, st));
                    if (prec_dbg == 3) HIP_TRY(hipDeviceSynchronize());
                    if (prec_dbg == 4) HIP_TRY(hipStreamSynchronize(st));        // lab: host-side order after it, other stream keeps running
                }
                else HIP_TRY(launch_attention(*c->tune, dt, sl.QKV, sl.U, n, N, D, c->H, st));
            }
        }
#ifdef VITX_LAB
        if (il == 0 && getenv("VITX_SNAP")) {      // lab: QKV (hi plane) and the attention output of layer 0 of this slice, to files (synchronous)
            static int run[4] = {0, 0, 0, 0};
            const int si = (int)(&sl - &c->slices[0]);
            std::vector<char> hq((size_t)M_real * 3 * D * 2), hu((size_t)M_real * D * 2);
            HIP_TRY(hipMemcpyAsync(hq.data(), sl.QKV, hq.size(), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(hu.data(), sl.U, hu.size(), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            char fn[512];
            snprintf(fn, sizeof fn, "%s/snap_s%d_r%d_qkv.bin", getenv("VITX_SNAP"), si, run[si]); { FILE *f = fopen(fn, "wb"); if (f) { fwrite(hq.data(), 1, hq.size(), f); fclose(f); } }
            snprintf(fn, sizeof fn, "%s/snap_s%d_r%d_u.bin", getenv("VITX_SNAP"), si, run[si]); { FILE *f = fopen(fn, "wb"); if (f) { fwrite(hu.data(), 1, hu.size(), f); fclose(f); } }
            ++run[si];
        }
#endif
        // output projection + residual (vit.cpp:868-873), then norm2 (vit.cpp:881-885) -> U2
        if ((rc = resid_gemm_ln(PC_GEMM_PROJ, sl.U, Wl[W_PROJ], w.proj_b, D, Fl[W_PROJ], w.ln2_w, w.ln2_b, sl.U2, &fix_u2))) return rc;
        // MLP (vit.cpp:889-900), then the NEXT layer's norm1 (vit.cpp:808-812) -> U; the last layer is followed by the cls-row norm instead
        if ((rc = gemm(c, tn_, st, PC_GEMM_FC1, EPI_BIAS_GELU, sl.U2, Wl[W_FC1], w.fc1_b, sl.Hbuf, nullptr, M, M_real, 4 * D, round_up(4 * D, tn), D, D, D, 4 * D, 0, 2, Fl[W_FC1], nullptr,
                       fix_u2.todo ? &fix_u2 : nullptr))) return rc;
        const LayerW *nx = il + 1 < c->L ? &c->layers[il + 1] : nullptr;
        if ((rc = resid_gemm_ln(PC_GEMM_FC2, sl.Hbuf, Wl[W_FC2], w.fc2_b, 4 * D, Fl[W_FC2], nx ? nx->ln1_w : nullptr, nx ? nx->ln1_b : nullptr, sl.U, &fix_u))) return rc;
        if (!c->trace_ids.empty() && (rc = trace(il + 1))) return rc;
    }
    // cls pooling + final norm (vit.cpp:910-919): row b*N of X, i.e. row stride N*D.  ViTSTR (vitstr.cpp:864-895) keeps the first
    // R = 25 tokens of every image instead: output row r = image r / R, token r % R.
    const int nR = n * c->R;
    {
        ProfScope ps(c, st, PC_LAYERNORM, 0, (double)nR * D * (4 + eb));
        // classifier: one row per image, row stride N*D; ViTSTR: groups of R consecutive token rows (stride D), group stride N*D
        if (tail) HIP_TRY(launch_layernorm(dt, sl.Xc, D, c->norm_w, c->norm_b, sl.Z, D, n, D, c->hp.eps, st));
        else HIP_TRY(launch_layernorm(dt, sl.X, c->R == 1 ? (long)N * D : (long)D, c->norm_w, c->norm_b, sl.Z, D, nR, D, c->hp.eps, st, c->R, (long)N * D));
    }
    // classifier (vit.cpp:927-928) and class softmax (vit.cpp:931-933)
    float *lg = d_logits ? (float *)d_logits : sl.logits;
    const int ldl = d_logits ? c->C : c->C_pad;
    const void *head_w = c->head_w; const QuantW *head_f = nullptr;
    if (c->head_q.blocks) {
        if (fused_ok(c->head_q, round_up(nR, tm))) head_f = &c->head_q;
        else { const QuantW *todo[1] = {&c->head_q}; void *dst[1] = {sl.Wq_head}; if ((rc = expand(todo, dst, 1))) return rc; head_w = sl.Wq_head; }
    }
    if ((rc = gemm(c, tn_, st, PC_GEMM_HEAD, EPI_BIAS_F32, sl.Z, head_w, c->head_b, lg, nullptr, round_up(nR, tm), nR, c->C, c->C_pad, D, D, D, ldl, 0, 4, head_f))) return rc;
    {
        ProfScope ps(c, st, PC_SOFTMAX, 0, (double)nR * c->C * 8);
        HIP_TRY(launch_softmax(dt, lg, (float *)d_probs, nR, c->C, ldl, st));
    }
    if (fuse && c->ln_fb_host) HIP_TRY(hipMemcpyAsync(c->ln_fb_host + (&sl - &c->slices[0]), sl.ln_todo + sl.ln_blocks, sizeof(unsigned), hipMemcpyDeviceToHost, st));       // fall-back budget
    return VITX_OK;
}

// Sub-batch sizes.  One 256x256 GEMM tile per CU per round means a sub-batch is cheapest when its tile counts land just
// under whole rounds: for ViT-B on 256 CUs 110 images are 85 row tiles = 255 / 765 / 1020 tiles for N = 768 / 2304 / 3072
// (1, 3 and 4 rounds) while 128 images cost 2 rounds' worth of time for 1.15 rounds of proj / fc2 work.  The first
// sub-batch size is the minimiser of a tile-round model of the four GEMMs of a layer (same tiling rules as launch_gemm);
// VITX_SPLIT=<images> overrides it.  More than two sub-batches are split evenly.
static double gemm_round_cost(long rows, int N, int K, int n_cu) {
    const long ntm = (rows + 255) / 256, ntn = (N + 255) / 256, tiles = ntm * ntn, rounds = tiles / n_cu, rem = tiles % n_cu;
    const double slots = K / 32.0, t_tile = slots * 0.98 + 3.5, t_half = slots * 0.6 + 3.0;
    if (rounds >= 1 && rem > 0 && rem <= n_cu * 6 / 10) {
        const long m_main = rounds * n_cu / ntn, half_tiles = ((ntm - m_main) * 2) * ntn;
        return rounds * t_tile + (double)((half_tiles + n_cu - 1) / n_cu) * t_half;
    }
    if (tiles < 128) return (double)(((rows + 127) / 128 * ntn + n_cu - 1) / n_cu) * t_half;
    return (double)((tiles + n_cu - 1) / n_cu) * t_tile;
}
// the LayerNorm-fusing residual GEMMs (proj, fc2) run rounds of gemm_ln_grid() workgroups: whole row blocks per XCD, column tiles of a
// row block in the same round; + 2 units per tile for the statistics exchange and the normalised store
static double gemm_round_cost_ln(long rows, int N, int K, int n_cu) {
    const long ntm = (rows + 255) / 256, ntn = N / 256;
    const int grid = gemm_ln_grid(n_cu, (int)(ntm * 256), N);
    const long most = ((ntm + 7) / 8) * ntn, wgx = grid / 8;
    return (double)((most + wgx - 1) / wgx) * (K / 32.0 * 0.98 + 3.5 + 2.0);
}
static void split_batch(const vitx_ctx *c, int n, int ns, int *m) {
    const int base = n / ns, extra = n % ns;
    for (int i = 0; i < ns; ++i) m[i] = base + (i < extra ? 1 : 0);
    if (c->split_first > 0 && ns == 2 && c->split_first < n) { m[0] = c->split_first; m[1] = n - c->split_first; return; }
    if (ns != 2) return;
    const int n_cu = c->tune->n_cu;
    const int D = c->D;
    const bool ln = c->ln_fuse && c->slices[0].ln_sync;
    if (ln) {
        // LayerNorm-fusing residual GEMMs run rounds of (CUs / 8 / ntn) * ntn workgroups per XCD: the first sub-batch takes as many images as
        // ONE such round holds (ViT-B on 256 CUs: 10 row blocks per XCD = 80 blocks = 103 images), the second the rest -- measured (r03a,
        // interleaved, ms per 256-image forward): 72 | 103 | model's 110 | 128 images first = 9.95 | 9.90 | 10.09 | 10.21
        const int ntn = D / 256, per_xcd = std::max(n_cu / 8, ntn) / ntn;
        const int s1 = (int)((long)8 * per_xcd * 256 / c->N);
        if (s1 >= n / 4 && s1 <= n / 2 && (long)s1 * c->N / 256 * ntn >= 128) { m[0] = s1; m[1] = n - s1; return; }
    }
    auto layer = [&](int imgs) {
        const long rows = (long)imgs * c->N;
        const bool fl = ln && ((rows + 255) / 256) * (D / 256) >= 128;        // the fused kernel needs the wide path (is_wide, kernels.hip)
        return gemm_round_cost(rows, 3 * D, D, n_cu) + gemm_round_cost(rows, 4 * D, D, n_cu) +
               (fl ? gemm_round_cost_ln(rows, D, D, n_cu) + gemm_round_cost_ln(rows, D, 4 * D, n_cu) : gemm_round_cost(rows, D, D, n_cu) + gemm_round_cost(rows, D, 4 * D, n_cu));
    };
    double best = layer(m[0]) + layer(m[1]);
    for (int s1 = std::max(8, n / 4); s1 <= n / 2; ++s1) {
        const double cost = layer(s1) + layer(n - s1);
        if (cost < best * 0.97) { best = cost; m[0] = s1; m[1] = n - s1; }   // move off the even split only for a clear (>3 %) modelled gain
    }
}

// Replay (or, the second time a call repeats, capture) the single-stream forward as a hipGraph.  *done = the forward was enqueued
// through a graph; otherwise the caller launches it directly (unless an error is returned).
static int forward_graph(vitx_ctx *c, hipStream_t st, const void *d_imgs, int n, void *d_probs, void *d_logits, bool *done) {
    *done = false;
    for (auto &ge : c->graphs)
        if (ge.imgs == d_imgs && ge.n == n && ge.probs == d_probs && ge.logits == d_logits) {
            if (hipGraphLaunch(ge.exec, st) == hipSuccess) { *done = true; return VITX_OK; }
            (void)hipGetLastError(); c->graphs_on = false; return VITX_OK;       // never seen; stay on the direct path from here on
        }
    vitx_ctx::GraphEntry &last = c->graph_last;
    const bool repeat = last.imgs == d_imgs && last.n == n && last.probs == d_probs && last.logits == d_logits;
    last = vitx_ctx::GraphEntry{d_imgs, d_probs, d_logits, n, nullptr};
    if (!repeat) return VITX_OK;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return VITX_OK; }   // the caller is capturing: our launches join ITS graph
    if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); c->graphs_on = false; return VITX_OK; }
    const int rc = forward_slice(c, c->slices[0], st, d_imgs, 0, n, d_probs, d_logits);
    hipGraph_t g = nullptr;
    const hipError_t e = hipStreamEndCapture(st, &g);
    hipGraphExec_t exec = nullptr;
    if (rc != VITX_OK || e != hipSuccess || !g || hipGraphInstantiate(&exec, g, nullptr, nullptr, 0) != hipSuccess) {
        if (g) (void)hipGraphDestroy(g);
        (void)hipGetLastError(); c->graphs_on = false;
        return rc;                               // nothing ran: a launch error is reported, a capture problem falls back to direct launches
    }
    (void)hipGraphDestroy(g);
    if (c->graphs.size() >= 8) { (void)hipGraphExecDestroy(c->graphs.front().exec); c->graphs.erase(c->graphs.begin()); }
    c->graphs.push_back(vitx_ctx::GraphEntry{d_imgs, d_probs, d_logits, n, exec});
    if (hipGraphLaunch(exec, st) != hipSuccess) { set_error("vitx_forward_device: hipGraphLaunch: %s", hipGetErrorString(hipGetLastError())); return VITX_ERR_HIP; }
    *done = true;
    return VITX_OK;
}

// Do the internal sub-batch streams really run BESIDE the caller's stream?  The HIP runtime maps streams onto a pool of hardware queues and
// the queues onto the command processor's slots; which streams end up serialised depends on every other stream alive in the process
// (measured r03, tools/ctx_order_probe.py: with earlier contexts still alive the 2nd and the 7th context of a process ran 11.6 instead of
// 9.9 ms per forward -- with 8 hardware queues the 2nd, 4th and 6th, with 2 none; per-kernel times unchanged).  Nothing in the API tells,
// so it is measured: a 40 us do-nothing kernel on each stream, forked and joined like a forward; ~40 us = concurrent, ~80 us = serialised.
// For a serialised internal stream up to 8 candidate streams are created and kept alive TOGETHER (a stream created after another was
// destroyed gets the same queue back), the first one that runs beside the caller's stream is adopted, the rest are destroyed.
// Once per context (on the first forward of >= 16 images: the first caller stream it sees), ~0.2 ms, synchronous -- documented in vitx.h;
// skipped while the caller captures a graph.
static int probe_pair(vitx_ctx *c, hipStream_t st, hipStream_t s1, hipEvent_t done, float *best_ms) {
    *best_ms = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {        // the first pass also wakes the queues up
        HIP_TRY(hipEventRecord(c->probe_a, st));
        HIP_TRY(hipStreamWaitEvent(s1, c->probe_a, 0));
        HIP_TRY(launch_spin(40, s1));
        HIP_TRY(hipEventRecord(done, s1));
        HIP_TRY(launch_spin(40, st));
        HIP_TRY(hipStreamWaitEvent(st, done, 0));
        HIP_TRY(hipEventRecord(c->probe_b, st));
        HIP_TRY(hipEventSynchronize(c->probe_b));
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, c->probe_a, c->probe_b));
        *best_ms = std::min(*best_ms, ms);
    }
    return VITX_OK;
}
static int ensure_concurrent(vitx_ctx *c, hipStream_t st, int ns) {
    if (ns < 2 || std::find(c->probed_streams.begin(), c->probed_streams.end(), st) != c->probed_streams.end()) return VITX_OK;
    // Only the FIRST caller stream a context sees is probed (and may get the internal streams replaced): a caller that alternates streams
    // must not pay a synchronising probe per call, and replacing an internal stream for the second caller stream could undo what was
    // found for the first (r03 advisor).  Later caller streams are remembered and left alone.
    if (!c->probed_streams.empty()) { if (c->probed_streams.size() < 16) c->probed_streams.push_back(st); return VITX_OK; }
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return VITX_OK; }
    if (!c->probe_a) { HIP_TRY(hipEventCreate(&c->probe_a)); HIP_TRY(hipEventCreate(&c->probe_b)); }
    int least = 0, greatest = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
    const float kSerialised = 0.064f;
    for (int i = 1; i < ns; ++i) {
        vitx_ctx::Slice &sl = c->slices[i];
        float ms = 0.0f;
        int rc = probe_pair(c, st, sl.stream, sl.done, &ms);
        if (rc) return rc;
        if (ms < kSerialised) continue;
        hipStream_t cand[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        int pick = -1;
        for (int k = 0; k < 8 && pick < 0; ++k) {
            HIP_TRY(hipStreamCreateWithPriority(&cand[k], hipStreamNonBlocking, (k & 1) ? 0 : greatest));
            ++c->stream_retries;
            if ((rc = probe_pair(c, st, cand[k], sl.done, &ms))) break;
            if (ms < kSerialised) pick = k;
        }
        for (int k = 0; k < 8; ++k) if (cand[k] && k != pick) (void)hipStreamDestroy(cand[k]);
        if (rc) return rc;
        if (pick >= 0) { (void)hipStreamDestroy(sl.stream); sl.stream = cand[pick]; }       // otherwise keep the original: nothing better exists
    }
    c->probed_streams.push_back(st);
    return VITX_OK;
}

static int forward_pass(vitx_ctx *c, const void *d_imgs, int n, void *d_probs, void *d_logits, hipStream_t st);
int vitx_forward_device(vitx_ctx *c, const void *d_imgs, int n, void *d_probs, void *d_logits, void *stream) {
    if (!c || !d_imgs || !d_probs) { set_error("vitx_forward_device: NULL argument"); return VITX_ERR_ARG; }
    if (n <= 0 || n > c->max_batch) { set_error("vitx_forward_device: batch %d outside 1..%d", n, c->max_batch); return VITX_ERR_ARG; }
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    // one pass of the kernels takes call_limit images (32-bit buffer window, vitx_ctx_create_ex); a larger batch is several passes, back to back on the
    // caller's stream through the same scratch -- images are independent, so the results are the ones a single pass would give
    if (n > c->call_limit && !c->trace_ids.empty()) { set_error("vitx_forward_device: the residual-stream trace takes one pass (at most %d images)", c->call_limit); return VITX_ERR_ARG; }
    for (int i0 = 0; i0 < n; i0 += c->call_limit) {
        const int ni = std::min(c->call_limit, n - i0);
        const int rc = forward_pass(c, (const float *)d_imgs + (size_t)i0 * c->S * c->S * c->Cin, ni, (float *)d_probs + (size_t)i0 * c->R * c->C,
                                    d_logits ? (float *)d_logits + (size_t)i0 * c->R * c->C : nullptr, st);
        if (rc) return rc;
    }
    return VITX_OK;
}
static int forward_pass(vitx_ctx *c, const void *d_imgs, int n, void *d_probs, void *d_logits, hipStream_t st) {
    // Fall-back budget of the fused LayerNorm.  The counters are what the PREVIOUS forwards copied to pinned memory behind their last kernel
    // (no event between that copy and this read: a value that is one forward stale, or mid-update, moves the decision by one window at most --
    // the reads are volatile so that each is one 32-bit load).  Test mode: only with bit 4.
    if (c->ln_fb_host && (!c->ln_test || (c->ln_test & 4))) {
        constexpr int kLnWindow = 16, kLnBudget = 8;
        auto counters = [&]() { unsigned long long t = 0; for (int i = 0; i < c->nslices && i < 4; ++i) t += *(volatile unsigned *)(c->ln_fb_host + i); return t; };
        if (c->ln_fuse) {
            if (++c->ln_fb_forwards >= kLnWindow) {
                const unsigned long long total = counters();
                if (total - c->ln_fb_base > (unsigned long long)kLnWindow * kLnBudget) {
                    c->ln_fuse = false; c->ln_fuse_disabled = true;
                    c->ln_cool_left = c->ln_cool_len; c->ln_cool_len = std::min(c->ln_cool_len * 2, 1 << 16);
                }
                c->ln_fb_base = total; c->ln_fb_forwards = 0;
            }
        } else if (c->ln_fuse_disabled && --c->ln_cool_left <= 0) {       // cool-down over: try the fused path again (same bits either way)
            c->ln_fuse = true; c->ln_fuse_disabled = false;
            c->ln_fb_base = counters(); c->ln_fb_forwards = 0;
        }
    }
    // while per-kernel profiling is on, sub-batches run back to back on the caller's stream so that every
    // event pair brackets one kernel running alone (exclusive durations, comparable with rocprofv3 --stats)
    const bool serial = c->prof_on;
    const int ns = (c->nslices > 1 && n >= 8 * c->nslices) ? c->nslices : 1;
    if (ns == 1) {
        if (c->graphs_on && !c->prof_on && c->trace_ids.empty()) {
            bool done = false;
            const int rc = forward_graph(c, st, d_imgs, n, d_probs, d_logits, &done);
            if (rc != VITX_OK || done) return rc;
        }
        return forward_slice(c, c->slices[0], st, d_imgs, 0, n, d_probs, d_logits);
    }
    int m[4];
    split_batch(c, n, ns, m);
    if (!serial) { const int rc = ensure_concurrent(c, st, ns); if (rc) return rc; }
    // fork: slices 1.. wait for the caller's stream and run their contiguous sub-batches on the internal streams, slice 0 runs on the caller's
    // stream itself, which finally joins the others (slice 0 is enqueued LAST so that the host has already fed the other streams)
    if (!serial) HIP_TRY(hipEventRecord(c->fork, st));
    int off[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < ns; ++i) off[i + 1] = off[i] + m[i];
    for (int k = 0; k < ns; ++k) {
        const int i = serial ? k : (k + 1) % ns;           // 1, 2, .., 0
        vitx_ctx::Slice &sl = c->slices[i];
        hipStream_t ss = (serial || i == 0) ? st : sl.stream;
        if (!serial && i > 0) HIP_TRY(hipStreamWaitEvent(sl.stream, c->fork, 0));
        int rc = forward_slice(c, sl, ss, (const float *)d_imgs + (size_t)off[i] * c->S * c->S * c->Cin, off[i], m[i], (float *)d_probs + (size_t)off[i] * c->R * c->C,
                               d_logits ? (float *)d_logits + (size_t)off[i] * c->R * c->C : nullptr);
        if (rc) return rc;
        if (!serial && i > 0) HIP_TRY(hipEventRecord(sl.done, sl.stream));
    }
    if (!serial) for (int i = 1; i < ns; ++i) HIP_TRY(hipStreamWaitEvent(st, c->slices[i].done, 0));
    return VITX_OK;
}

int vitx_forward(vitx_ctx *c, const float *imgs, int n, float *probs, float *logits) {
    if (!c || !imgs || !probs) { set_error("vitx_forward: NULL argument"); return VITX_ERR_ARG; }
    if (n <= 0 || n > c->max_batch) { set_error("vitx_forward: batch %d outside 1..%d", n, c->max_batch); return VITX_ERR_ARG; }
    HIP_TRY(hipSetDevice(c->device));
    const size_t img_bytes = (size_t)n * c->S * c->S * c->Cin * 4;
    HIP_TRY(hipMemcpyAsync(c->img, imgs, img_bytes, hipMemcpyHostToDevice, c->stream));
    int rc = vitx_forward_device(c, c->img, n, c->probs, logits ? c->logits_all : nullptr, c->stream);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(probs, c->probs, (size_t)n * c->R * c->C * 4, hipMemcpyDeviceToHost, c->stream));
    if (logits) HIP_TRY(hipMemcpyAsync(logits, c->logits_all, (size_t)n * c->R * c->C * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return VITX_OK;
}

int vitx_ctx_synchronize(vitx_ctx *c) {
    if (!c) return VITX_ERR_ARG;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return VITX_OK;
}

int vitx_profile_enable(vitx_ctx *c, int on) {
    if (!c) return VITX_ERR_ARG;
    HIP_TRY(hipSetDevice(c->device));
    c->prof_on = on != 0; c->recs.clear(); c->ev_used = 0;
    if (on) {       // time origin for the busy-interval union (kernels of concurrent slices overlap)
        if (!c->prof_base) HIP_TRY(hipEventCreate(&c->prof_base));
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipEventRecord(c->prof_base, c->stream));
        HIP_TRY(hipEventSynchronize(c->prof_base));
    }
    return VITX_OK;
}

// What the HIP-event bracket of vitx_profile_enable adds to ONE launch: 32 x [record, 20 us kernel that stamps its own first and last
// wall-clock reading, record] queued back to back on the context's stream like a profiled forward; the median of
// (event interval - the kernel's own interval).  bench.py subtracts it from every launch of the profiled step (r04: the bracket read
// 5.3-5.5 us above the device's dispatch stamps for every kernel class, so `roofline.achieved` was 4-13 % low).
int vitx_profile_bracket_us(vitx_ctx *c, double *bracket_us) {
    if (!c || !bracket_us) return VITX_ERR_ARG;
    HIP_TRY(hipSetDevice(c->device));
    constexpr int NB = 32;
    long long *d_st = nullptr; hipEvent_t ev[2 * NB];
    HIP_TRY(hipMalloc((void **)&d_st, sizeof(long long) * 2 * NB));
    int made = 0; hipError_t e = hipSuccess;
    while (made < 2 * NB) { if ((e = hipEventCreate(&ev[made])) != hipSuccess) break; ++made; }      // `made` counts the handles that exist
    // nothing else of this device may be in flight (another slice stream's forward would stretch the brackets): the whole device, not only c->stream
    if (e == hipSuccess) e = hipDeviceSynchronize();
    for (int i = 0; i < NB && e == hipSuccess; ++i) {
        e = hipEventRecord(ev[2 * i], c->stream);
        if (e == hipSuccess) e = launch_spin_stamp(20, d_st + 2 * i, c->stream);
        if (e == hipSuccess) e = hipEventRecord(ev[2 * i + 1], c->stream);
    }
    long long h_st[2 * NB];
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(h_st, d_st, sizeof h_st, hipMemcpyDeviceToHost);
    std::vector<double> over;
    for (int i = 0; i < NB && e == hipSuccess; ++i) {
        float ms = 0.0f;
        e = hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]);
        over.push_back((double)ms * 1e3 - (double)(h_st[2 * i + 1] - h_st[2 * i]) * 0.01);
    }
    for (int i = 0; i < made; ++i) (void)hipEventDestroy(ev[i]);
    (void)hipFree(d_st);
    if (e != hipSuccess) { set_error("vitx_profile_bracket_us: %s", hipGetErrorString(e)); return VITX_ERR_HIP; }
    std::sort(over.begin(), over.end());
    *bracket_us = over[over.size() / 2];
    return VITX_OK;
}

int vitx_profile_read(vitx_ctx *c, vitx_prof_entry *out, int max_entries, int *n_entries) {
    if (!c || !out || !n_entries) return VITX_ERR_ARG;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());
    vitx_prof_entry acc[PC_COUNT];
    std::vector<std::pair<float, float>> iv[PC_COUNT];
    for (int i = 0; i < PC_COUNT; ++i) acc[i] = vitx_prof_entry{kProfNames[i], 0, 0.0, 0.0, 0.0, 0.0};
    for (const auto &r : c->recs) {
        float ms = 0.0f, ta = 0.0f, tb = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, r.a, r.b));
        if (c->prof_base) { HIP_TRY(hipEventElapsedTime(&ta, c->prof_base, r.a)); HIP_TRY(hipEventElapsedTime(&tb, c->prof_base, r.b)); iv[r.cls].push_back({ta, tb}); }
        acc[r.cls].launches++; acc[r.cls].total_ms += ms; acc[r.cls].flops += r.flops; acc[r.cls].bytes += r.bytes;
    }
    for (int i = 0; i < PC_COUNT; ++i) {        // union of this class's [start, stop] intervals over all streams
        std::sort(iv[i].begin(), iv[i].end());
        double busy = 0.0; float cur_a = 0, cur_b = -1;
        for (auto &p : iv[i]) {
            if (cur_b < cur_a || p.first > cur_b) { if (cur_b >= cur_a) busy += cur_b - cur_a; cur_a = p.first; cur_b = p.second; }
            else cur_b = std::max(cur_b, p.second);
        }
        if (cur_b >= cur_a && !iv[i].empty()) busy += cur_b - cur_a;
        acc[i].busy_ms = iv[i].empty() ? acc[i].total_ms : busy;
    }
    int k = 0;
    for (int i = 0; i < PC_COUNT && k < max_entries; ++i) if (acc[i].launches) out[k++] = acc[i];
    *n_entries = k;
    c->recs.clear(); c->ev_used = 0;
    return VITX_OK;
}

// ---- single-kernel entry points --------------------------------------------------------------
int vitx_op_layernorm(int dtype, const void *x, const void *w, const void *b, void *y, int M, int D, float eps, void *stream) {
    if (!x || !w || !b || !y || M <= 0) return VITX_ERR_ARG;
    hipError_t e = launch_layernorm(dtype, (const float *)x, D, (const float *)w, (const float *)b, y, D, M, D, eps, (hipStream_t)stream);
    if (e != hipSuccess) { set_error("vitx_op_layernorm: %s", hipGetErrorString(e)); return e == hipErrorInvalidValue ? VITX_ERR_UNSUPPORTED : VITX_ERR_HIP; }
    return VITX_OK;
}
static int op_gemm_impl(int dtype, int epi, int kernel, const void *a, const void *w, const void *bias, void *out, const void *pos, int M, int M_real, int N, int n_pad, int K, int tpi, void *stream) {
    if (!a || !w || !out || !bias || epi < 0 || epi > EPI_BIAS_HILO || M_real <= 0 || M_real > M || (epi == EPI_PATCH && (!pos || tpi <= 0))) { set_error("vitx_op_gemm_ex: invalid argument"); return VITX_ERR_ARG; }
    if (M % 128 || N % 4 || K % 64) { set_error("vitx_op_gemm: M %% 128, N %% 4, K %% 64 must be 0"); return VITX_ERR_ARG; }
    const Tuning *t0 = tuning_for_device(-1);
    if (!t0) { set_error("vitx_op_gemm: kernel bring-up failed"); return VITX_ERR_HIP; }
    Tuning t = *t0;
    if (kernel == 2) t.gemm_split = 1;
    else if (kernel == 1) t.gemm_cfg = 1;                  // ping-pong persistent kernel
    else if (kernel != 0) t.gemm_cfg = kernel;
    // W (and bias) must hold n_pad rows; rows beyond N are never stored
    GemmArgs g{};
    g.A = a; g.W = w; g.bias = (const float *)bias; g.out = out; g.pos = (const float *)pos;
    g.M = M; g.M_real = M_real; g.N = N; g.N_pad = n_pad; g.K = K; g.lda = K; g.ldw = K; g.ldo = N; g.tpi = tpi;
    g.hilo_off = epi == EPI_BIAS_HILO ? (long)M * N : 0;          // the lo plane follows the [M][N] hi plane
    hipError_t e = launch_gemm(t, dtype, epi, g, (hipStream_t)stream);
    if (e == hipErrorInvalidValue) { set_error("vitx_op_gemm: kernel %d cannot tile M %d N %d K %d", kernel, M, N, K); return VITX_ERR_UNSUPPORTED; }
    if (e != hipSuccess) { set_error("vitx_op_gemm: %s", hipGetErrorString(e)); return VITX_ERR_HIP; }
    return VITX_OK;
}
int vitx_op_gemm_ex(int dtype, int epi, int kernel, const void *a, const void *w, const void *bias, void *out, const void *pos, int M, int M_real, int N, int K, int tpi, void *stream) {
    return op_gemm_impl(dtype, epi, kernel, a, w, bias, out, pos, M, M_real, N, round_up(N, 256), K, tpi, stream);     // W and bias hold N rounded up to 256 rows
}
int vitx_op_gemm(int dtype, int epi, const void *a, const void *w, const void *bias, void *out, int M, int N, int K, void *stream) {
    if (epi < 0 || epi > 3 || N % 64) { set_error("vitx_op_gemm: epi 0..3, N %% 64 == 0"); return VITX_ERR_ARG; }
    return op_gemm_impl(dtype, epi, 0, a, w, bias, out, nullptr, M, M, N, round_up(N, gemm_tile_n()), K, 0, stream);   // W and bias hold N rounded up to 128 rows
}
// x[M][N] f32 += A[M][K] . W[N][K]^T + bias, then y[M][N] (dtype) = LayerNorm(x) * ln_w + ln_b computed by the GEMM's own epilogue
// (GemmLn) + the fix-up launch.  `test`: GemmLn::test (forced time-outs).  Synchronous; *fallbacks = tiles that took the fix-up path.
int vitx_op_gemm_ln(int dtype, const void *a, const void *w, const void *bias, void *x, const void *ln_w, const void *ln_b, void *y, int M, int N, int K, float eps,
                    int test, int timeout_us, int *fallbacks, void *stream) {
    if (!a || !w || !bias || !x || !ln_w || !ln_b || !y || M <= 0 || timeout_us < 0) { set_error("vitx_op_gemm_ln: invalid argument"); return VITX_ERR_ARG; }
    const Tuning *t0 = tuning_for_device(-1);
    if (!t0) { set_error("vitx_op_gemm_ln: kernel bring-up failed"); return VITX_ERR_HIP; }
    GemmArgs g{};
    g.A = a; g.W = w; g.bias = (const float *)bias; g.out = x; g.M = M; g.M_real = M; g.N = N; g.N_pad = N; g.K = K; g.lda = K; g.ldw = K; g.ldo = N;
    if (!gemm_ln_fusable(*t0, g)) { set_error("vitx_op_gemm_ln: M %d N %d K %d does not take the LayerNorm-fusing kernel (M %% 256, N in {256,512,768,1024}, >= 128 tiles, K %% 128)", M, N, K); return VITX_ERR_UNSUPPORTED; }
    static unsigned epoch = 0x40000000u;       // its own tag range (the scratch is private to the call anyway)
    const size_t nb = (size_t)M / 256, sync_bytes = nb * (N / 256) * 256 * 2 * sizeof(unsigned long long);
    unsigned long long *sync = nullptr; unsigned *todo = nullptr;
    HIP_TRY(hipMalloc((void **)&sync, sync_bytes));
    if (hipMalloc((void **)&todo, (nb + 1) * 4) != hipSuccess) { (void)hipFree(sync); return VITX_ERR_NOMEM; }
    int rc = VITX_OK;
    hipStream_t st = (hipStream_t)stream;
    GemmLn ln{};
    ln.w = (const float *)ln_w; ln.b = (const float *)ln_b; ln.out = y; ln.eps = eps; ln.sync = sync; ln.todo = todo; ln.fallbacks = todo + nb;
    ln.epoch = ++epoch; ln.timeout = (unsigned)timeout_us * 100u; ln.test = test;
    g.ln = &ln;
    hipError_t e = hipMemsetAsync(sync, 0, sync_bytes, st);
    if (e == hipSuccess) e = hipMemsetAsync(todo, 0, (nb + 1) * 4, st);
    if (e == hipSuccess) e = launch_gemm(*t0, dtype, EPI_BIAS_RESID, g, st);
    if (e == hipSuccess) e = launch_layernorm_fixup(dtype, (const float *)x, ln.w, ln.b, y, M, N, eps, todo, ln.epoch, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    unsigned fb = 0;
    if (e == hipSuccess) e = hipMemcpy(&fb, todo + nb, 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess) { set_error("vitx_op_gemm_ln: %s", hipGetErrorString(e)); rc = VITX_ERR_HIP; }
    if (fallbacks) *fallbacks = (int)fb;
    (void)hipFree(sync); (void)hipFree(todo);
    return rc;
}
// quantised-weight kernels (quant.hip, gemm_nt_kernel<.., Q4>): blocks in the FILE's byte layout for every type except q4_0, whose
// nibble plane / scale plane split is done here the way the context does it at upload
int vitx_op_dequant(int dtype, int qtype, const void *blocks, const void *scales, void *out, int N, int n_pad, int K, void *stream) {
    if (!blocks || !out || N <= 0 || n_pad < N || K <= 0 || K % 32 || (dtype != VITX_F16 && dtype != VITX_BF16)) { set_error("vitx_op_dequant: invalid argument"); return VITX_ERR_ARG; }
    DequantJob j{blocks, scales, out, N, n_pad, K / 32};
    hipError_t e = launch_dequant(dtype, qtype, &j, 1, (hipStream_t)stream);
    if (e != hipSuccess) { set_error("vitx_op_dequant: %s", hipGetErrorString(e)); return e == hipErrorInvalidValue ? VITX_ERR_ARG : VITX_ERR_HIP; }
    return VITX_OK;
}
int vitx_op_gemm_q4(int dtype, int epi, const void *a, const void *qs, const void *scales, const void *bias, void *out, int M, int M_real, int N, int K, void *stream) {
    if (!a || !qs || !scales || !bias || !out || epi < 0 || epi > EPI_BIAS_F32 || M_real <= 0 || M_real > M) { set_error("vitx_op_gemm_q4: invalid argument"); return VITX_ERR_ARG; }
    if (!tuning_for_device(-1)) { set_error("vitx_op_gemm_q4: kernel bring-up failed"); return VITX_ERR_HIP; }
    GemmArgs g{};
    g.A = a; g.W = qs; g.Wscale = (const uint16_t *)scales; g.bias = (const float *)bias; g.out = out;
    g.M = M; g.M_real = M_real; g.N = N; g.N_pad = round_up(N, 128); g.K = K; g.lda = K; g.ldw = K; g.ldo = N;
    hipError_t e = launch_gemm_q4(dtype, epi, g, (hipStream_t)stream);
    if (e == hipErrorInvalidValue) { set_error("vitx_op_gemm_q4: M %% 128, K %% 64 must be 0 (M %d N %d K %d)", M, N, K); return VITX_ERR_UNSUPPORTED; }
    if (e != hipSuccess) { set_error("vitx_op_gemm_q4: %s", hipGetErrorString(e)); return VITX_ERR_HIP; }
    return VITX_OK;
}
size_t vitx_ctx_weight_bytes(const vitx_ctx *c) { return c ? c->weight_bytes : 0; }
int vitx_ctx_shares_weights(const vitx_ctx *c) { return c && c->weights_shared ? 1 : 0; }
int vitx_ctx_stream_retries(const vitx_ctx *c) { return c ? c->stream_retries : -1; }
int vitx_ctx_ln_fusion_active(const vitx_ctx *c) { return c ? ((c->ln_fuse && !c->slices.empty() && c->slices[0].ln_sync && c->tune->n_xcd == 8) ? 1 : (c->ln_fuse_disabled ? -1 : 0)) : 0; }
long long vitx_ctx_ln_fallbacks(vitx_ctx *c) {
    if (!c) return -1;
    if (hipSetDevice(c->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return -1;
    long long total = 0;
    for (auto &sl : c->slices) {
        if (!sl.ln_todo) continue;
        unsigned v = 0;
        if (hipMemcpy(&v, sl.ln_todo + sl.ln_blocks, sizeof v, hipMemcpyDeviceToHost) != hipSuccess) return -1;
        total += v;
    }
    return total;
}

int vitx_op_attention_ex(int dtype, int kernel, const void *qkv, void *out, int n_img, int N, int D, int H, void *stream) {
    if (!qkv || !out || n_img <= 0 || kernel < 0) return VITX_ERR_ARG;
    const Tuning *t0 = tuning_for_device(-1);
    if (!t0) { set_error("vitx_op_attention: kernel bring-up failed"); return VITX_ERR_HIP; }
    Tuning t = *t0;
#ifdef VITX_LAB
    t.attn_flags = kernel >> 4; kernel &= 15;                // bits 4+: ablation build of the pipelined kernel (tools/attn_bench.py only)
#endif
    if (kernel == ATTN_PERSIST) {                            // persistent single-pass kernel (193..224 tokens)
        if (N <= 192 || N > 224) { set_error("vitx_op_attention: the persistent kernel takes 193..224 tokens, not %d", N); return VITX_ERR_UNSUPPORTED; }
    } else if (kernel == ATTN_SINGLE) {                      // single-pass kernel, also where the automatic choice prefers the pipelined one
        if (!attention_single_pass_supports(N)) { set_error("vitx_op_attention: no single-pass instantiation for %d tokens", N); return VITX_ERR_UNSUPPORTED; }
    } else if (kernel == ATTN_STREAM) {                      // streaming two-pass kernel (attention_stream.hip), head dim 64
        if (!attention_stream_supports(n_img, N, D, H)) { set_error("vitx_op_attention: the streaming kernel needs head_dim 64"); return VITX_ERR_UNSUPPORTED; }
    } else if (kernel != ATTN_AUTO && kernel != ATTN_FLOW) { set_error("vitx_op_attention: unknown kernel id %d", kernel); return VITX_ERR_ARG; }
    t.attn_kernel = kernel;
    hipError_t e = launch_attention(t, dtype, qkv, out, n_img, N, D, H, (hipStream_t)stream);
    if (e != hipSuccess) { set_error("vitx_op_attention: %s", hipGetErrorString(e)); return e == hipErrorInvalidValue ? VITX_ERR_UNSUPPORTED : VITX_ERR_HIP; }
    return VITX_OK;
}
int vitx_op_attention(int dtype, const void *qkv, void *out, int n_img, int N, int D, int H, void *stream) { return vitx_op_attention_ex(dtype, 0, qkv, out, n_img, N, D, H, stream); }
// The precise kernel on planes that are already split (what the QKV GEMM's epilogue 5 emits): d_hi [n_img * N][3 D] fp16, the lo plane lo_off
// ELEMENTS behind it.  Only enqueues on `stream`.
int vitx_op_attention_planes(const void *d_hi, long lo_off, void *out, int n_img, int N, int D, int H, void *stream) {
    if (!d_hi || !out || n_img <= 0 || N <= 0 || D <= 0 || H <= 0) { set_error("vitx_op_attention_planes: invalid argument"); return VITX_ERR_ARG; }
    // the lo plane lies a whole number of 4-element groups behind the hi plane's rows and inside the 32-bit byte window the kernels address
    if (lo_off < (long)n_img * N * 3 * D || lo_off % 4 != 0 || (size_t)lo_off * 2 + (size_t)n_img * N * 3 * D * 2 > 0xf0000000u) {
        set_error("vitx_op_attention_planes: lo_off %ld must be a multiple of 4 elements, at least n_img * N * 3 * D = %ld, and keep both planes below 0xf0000000 bytes", lo_off, (long)n_img * N * 3 * D);
        return VITX_ERR_ARG;
    }
    if (!tuning_for_device(-1)) { set_error("vitx_op_attention_planes: kernel bring-up failed"); return VITX_ERR_HIP; }
    if (!attention_stream_supports(n_img, N, D, H)) { set_error("vitx_op_attention_planes: head_dim must be 64"); return VITX_ERR_UNSUPPORTED; }
    hipError_t e = launch_attention_stream(DT_F16, true, d_hi, out, n_img, N, D, H, lo_off, (hipStream_t)stream);
    if (e != hipSuccess) { set_error("vitx_op_attention_planes: %s", hipGetErrorString(e)); return VITX_ERR_HIP; }
    return VITX_OK;
}
// Attention of token 0 of every image (the row the last layer of a classifier keeps, vit.cpp:910-911): out[n_img][D] (dtype).  lo_off != 0: the two
// fp16 planes of the F16 parity mode (VITX_F16 only), as vitx_op_attention_planes takes them.  Only enqueues on `stream`.
int vitx_op_attention_cls(int dtype, const void *d_qkv, long lo_off, void *out, int n_img, int N, int D, int H, void *stream) {
    if (!d_qkv || !out || n_img <= 0 || N <= 0 || D <= 0 || H <= 0 || (dtype != VITX_F16 && dtype != VITX_BF16)) { set_error("vitx_op_attention_cls: invalid argument"); return VITX_ERR_ARG; }
    if (lo_off && (dtype != VITX_F16 || lo_off < (long)n_img * N * 3 * D || lo_off % 8 != 0)) {
        set_error("vitx_op_attention_cls: a lo plane needs VITX_F16 and lo_off %ld a multiple of 8 elements, at least n_img * N * 3 * D = %ld", lo_off, (long)n_img * N * 3 * D);
        return VITX_ERR_ARG;
    }
    if (!attention_cls_supports(N, D, H)) { set_error("vitx_op_attention_cls: head_dim must be 8, 16, 32, 64 or 128 and N at most 15360 (head_dim %d, N %d)", D / H, N); return VITX_ERR_UNSUPPORTED; }
    hipError_t e = launch_attention_cls(dtype == VITX_F16 ? DT_F16 : DT_BF16, d_qkv, lo_off, out, nullptr, nullptr, n_img, N, D, H, (hipStream_t)stream);
    if (e != hipSuccess) { set_error("vitx_op_attention_cls: %s", hipGetErrorString(e)); return VITX_ERR_HIP; }
    return VITX_OK;
}
// The parity mode's attention on f32 q, k, v (what the reference multiplies, vit.cpp:848,858): splits the rows into the two fp16 planes the
// QKV GEMM's EPI_BIAS_HILO epilogue emits, then runs the precise streaming kernel.  Synchronous (allocates its own scratch).
int vitx_op_attention_f32(const float *qkv_f32, void *out, int n_img, int N, int D, int H, void *stream) {
    if (!qkv_f32 || !out || n_img <= 0 || N <= 0 || D <= 0 || H <= 0) { set_error("vitx_op_attention_f32: invalid argument"); return VITX_ERR_ARG; }
    if (!tuning_for_device(-1)) { set_error("vitx_op_attention_f32: kernel bring-up failed"); return VITX_ERR_HIP; }
    if (!attention_stream_supports(n_img, N, D, H)) { set_error("vitx_op_attention_f32: head_dim must be 64"); return VITX_ERR_UNSUPPORTED; }
    const size_t n = (size_t)n_img * N * 3 * D;
    void *planes = nullptr;
    HIP_TRY(hipMalloc(&planes, n * 2 * 2 + 64));
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = launch_split_hilo(DT_F16, qkv_f32, planes, (char *)planes + n * 2, n, st);
    if (e == hipSuccess) e = launch_attention_stream(DT_F16, true, planes, out, n_img, N, D, H, (long)n, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(planes);
    if (e != hipSuccess) { set_error("vitx_op_attention_f32: %s", hipGetErrorString(e)); return VITX_ERR_HIP; }
    return VITX_OK;
}
int vitx_op_softmax_dt(int dtype, const void *logits, void *probs, int rows, int cols, int ld, void *stream) {
    if (!logits || !probs || rows <= 0 || cols <= 0 || (dtype != VITX_F16 && dtype != VITX_BF16)) return VITX_ERR_ARG;
    hipError_t e = launch_softmax(dtype, (const float *)logits, (float *)probs, rows, cols, ld, (hipStream_t)stream);
    if (e != hipSuccess) { set_error("vitx_op_softmax: %s", hipGetErrorString(e)); return VITX_ERR_HIP; }
    return VITX_OK;
}
int vitx_op_softmax(const void *logits, void *probs, int rows, int cols, int ld, void *stream) { return vitx_op_softmax_dt(VITX_F16, logits, probs, rows, cols, ld, stream); }

int vitx_trace_enable(vitx_ctx *c, const int32_t *image_ids, int n) {
    if (!c || n < 0 || (n > 0 && !image_ids)) return VITX_ERR_ARG;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());
    if (c->trace_buf) { (void)hipFree(c->trace_buf); c->trace_buf = nullptr; }
    c->trace_ids.assign(image_ids, image_ids + n);
    for (int id : c->trace_ids) if (id < 0 || id >= c->max_batch) { c->trace_ids.clear(); set_error("vitx_trace_enable: image id %d outside 0..%d", id, c->max_batch - 1); return VITX_ERR_ARG; }
    if (n) HIP_TRY(hipMalloc((void **)&c->trace_buf, (size_t)(c->L + 1) * n * c->N * c->D * 4));
    return VITX_OK;
}
int vitx_trace_read(vitx_ctx *c, float *out, size_t n_floats) {
    if (!c || !out) return VITX_ERR_ARG;
    const size_t need = (size_t)(c->L + 1) * c->trace_ids.size() * c->N * c->D;
    if (!c->trace_buf || n_floats < need) { set_error("vitx_trace_read: trace not enabled or buffer too small (%zu floats needed)", need); return VITX_ERR_ARG; }
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, c->trace_buf, need * 4, hipMemcpyDeviceToHost));
    return VITX_OK;
}

int vitx_preprocess_u8_device(const void *d_hwc, int n, int nx, int ny, int img_size, int interp, void *d_out, void *stream) {
    if (!d_hwc || !d_out || n <= 0 || nx <= 0 || ny <= 0 || img_size <= 0) { set_error("vitx_preprocess_u8_device: invalid argument"); return VITX_ERR_ARG; }
    if (interp != VITX_BICUBIC && interp != VITX_BILINEAR) { set_error("vitx_preprocess_u8_device: interpolation mode %d is not supported", interp); return VITX_ERR_ARG; }
    hipError_t e = launch_preprocess(d_hwc, (float *)d_out, n, nx, ny, img_size, interp == VITX_BICUBIC, (hipStream_t)stream);
    if (e != hipSuccess) { set_error("vitx_preprocess_u8_device: %s", hipGetErrorString(e)); return VITX_ERR_HIP; }
    return VITX_OK;
}

}  // extern "C"

/*
 * vitx.h -- C ABI of the MI355X-native ViT forward engine (libvitx.so).
 *
 * This is the drop-in boundary for the forward path of staghado/vit.cpp.  The
 * reference exposes three C++ entry points (vit.h:119-122):
 *     bool vit_model_load(const std::string&, vit_model&);                      vit.h:120
 *     bool vit_image_preprocess(const image_u8&, image_f32&, const vit_hparams&); vit.h:119
 *     int  vit_predict(const vit_model&, vit_state&, const image_f32,
 *                      const vit_params&, std::vector<std::pair<float,int>>&);  vit.h:122
 * The C++ mirror of those signatures lives in vit.cpp_amd/vit.h and is a thin
 * wrapper over the functions below (plain pointers and sizes, int status codes,
 * caller-owned output buffers, no exceptions across the ABI).  INTEGRATION.md
 * shows the binding a maintainer of the reference would add.
 *
 * Threading: a vitx_model is immutable after load and may be shared; a vitx_ctx
 * is per (thread, GPU) mutable scratch -- the analogue of vit_state (vit.h:72-80).
 */
#ifndef VITX_H
#define VITX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vitx_model vitx_model;
typedef struct vitx_ctx vitx_ctx;

enum vitx_status {
    VITX_OK = 0,
    VITX_ERR_IO = 1,          /* cannot open / short read                         (vit.cpp:312-317) */
    VITX_ERR_FORMAT = 2,      /* bad magic, unknown tensor, wrong shape or size   (vit.cpp:320-328, 618-685, 697-701) */
    VITX_ERR_ARG = 3,         /* NULL pointer, batch > max_batch, wrong image size (vit.cpp:757) */
    VITX_ERR_HIP = 4,         /* a HIP runtime call or kernel launch failed */
    VITX_ERR_UNSUPPORTED = 5, /* model shape the kernels do not cover */
    VITX_ERR_NOMEM = 6
};

/* Arithmetic type of the MFMA operands (accumulation, LayerNorm, softmax and the
 * residual stream are always f32).  F16 reproduces the reference's rounding points
 * (ggml rounds mul_mat activations to fp16); BF16 is the mode BASELINE.json names. */
enum vitx_dtype { VITX_F16 = 0, VITX_BF16 = 1 };

/* Interpolation of vit_image_preprocess (vit_hparams::interpolation, vit.h:30). */
enum vitx_interp { VITX_BICUBIC = 0, VITX_BILINEAR = 1 };

/* Mirrors vit_hparams (vit.h:20-37); eps is not stored in the file (always 1e-6). */
typedef struct vitx_hparams {
    int32_t hidden_size;
    int32_t num_hidden_layers;
    int32_t num_attention_heads;
    int32_t num_classes;
    int32_t patch_size;
    int32_t img_size;
    int32_t ftype;
    float eps;
} vitx_hparams;

const char *vitx_status_str(int status);
/* Thread-local description of the last error raised on this thread ("" if none). */
const char *vitx_last_error(void);

/* ---- model file (replaces vit_model_load, vit.cpp:308-712) ------------------ */
/* Parses the legacy-ggml ".gguf" file into host memory; validates magic, names,
 * shapes and byte sizes exactly where the reference does.  No GPU is touched. */
int vitx_model_load(const char *path, vitx_model **out);
void vitx_model_free(vitx_model *m);
/* Unique id of this load within the process (> 0; 0 for NULL).  A context cache must key on this, not on the pointer: a freed model's
 * address is routinely handed to the next vitx_model_load. */
uint64_t vitx_model_uid(const vitx_model *m);
int vitx_model_hparams(const vitx_model *m, vitx_hparams *out);
int vitx_model_num_labels(const vitx_model *m);
/* id2label lookup (vit.cpp:1065 uses .at(idx)); NULL when the id has no label. */
const char *vitx_model_label(const vitx_model *m, int class_id);
/* 3 for a ViT classifier file; 1 for a ViTSTR scene-text file (extensions/vitstr.cpp: the patch kernel is [P, P, 1, D],
 * vitstr.cpp:482).  vitx_model_seq_len: 0 for a classifier (one probability row per image: the cls token), 25 for ViTSTR
 * (the head reads tokens 0..24 of every image, vitstr.cpp:864-904: 25 probability rows per image). */
#define VITX_VITSTR_SEQ_LEN 25
int vitx_model_in_channels(const vitx_model *m);
int vitx_model_seq_len(const vitx_model *m);
int vitx_model_num_tensors(const vitx_model *m);
/* Name, file type code (0 f32,1 f16,2 q4_0,3 q4_1,6 q5_0,7 q5_1,8 q8_0), ggml-order dims. */
int vitx_model_tensor_info(const vitx_model *m, int index, const char **name, int32_t *type, int64_t ne[4], size_t *nbytes);
/* Decodes tensor `index` to f32 into out (n_elements floats).  Host only. */
int vitx_model_tensor_f32(const vitx_model *m, int index, float *out, size_t n_elements);

/* Re-encodes an f16/f32 model file with its 2-D "*weight" tensors in block format `ftype`
 * (2 q4_0, 3 q4_1, 6 q5_0, 7 q5_1, 8 q8_0), byte-identical to what the reference's offline
 * `quantize` tool writes (quantize.cpp:34-353: header ftype, label order, which tensors, block
 * encoders).  Host only.  VITX_ERR_ARG for another ftype, VITX_ERR_FORMAT if already quantised. */
int vitx_quantize_file(const char *path_in, const char *path_out, int ftype);

/* ---- image files (replaces load_image_from_file = stbi_load(..., 3), vit.cpp:109-127) ---- */
/* Decodes a JPEG (baseline or progressive Huffman, 8-bit, gray or YCbCr), a non-interlaced PNG or a binary PPM into tightly
 * packed RGB u8 [ny][nx][3], top row first -- what stbi_load(fname, &nx, &ny, &nc, 3) hands the reference.  *out_rgb is
 * malloc'ed; release it with vitx_image_free.  VITX_ERR_IO if the file cannot be read, VITX_ERR_FORMAT if it cannot be decoded. */
int vitx_image_load(const char *path, uint8_t **out_rgb, int *nx, int *ny);
int vitx_image_decode(const uint8_t *bytes, size_t n_bytes, uint8_t **out_rgb, int *nx, int *ny);
void vitx_image_free(uint8_t *rgb);

/* ---- preprocess (replaces vit_image_preprocess, vit.cpp:289-305) ------------ */
/* u8 HWC RGB [ny][nx][3] -> f32 HWC [img_size][img_size][3], resized without
 * crop/antialias, rounded to u8, ImageNet mean/std normalised (vit.cpp:130-287). */
int vitx_preprocess_u8(const uint8_t *hwc, int nx, int ny, int img_size, int interp, float *out_hwc);

/* The same on the GPU for n images of one source size: d_hwc u8 [n][ny][nx][3] -> d_out f32
 * [n][img_size][img_size][3], both device pointers; only enqueues on `stream`.  Bit-identical
 * to vitx_preprocess_u8 (same operations in the same order, IEEE division, no FMA contraction). */
int vitx_preprocess_u8_device(const void *d_hwc, int n, int nx, int ny, int img_size, int interp, void *d_out_hwc, void *stream);

/* ViTSTR (extensions/vitstr.cpp) front and back end.  vitx_preprocess_vitstr_u8 replaces that extension's vit_image_preprocess
 * (vitstr.cpp:135-201): RGB u8 HWC -> grey (PIL weights, truncated to u8) -> direct linear resize to img_size^2 -> [-1, 1]; out is
 * ONE channel [img_size][img_size] f32 (nx, ny >= 2).  vitx_vitstr_decode replaces the greedy decode of its vit_predict
 * (vitstr.cpp:1025-1051) on one image's [seq_len][num_classes] probabilities: ids gets the classes of the characters (at most
 * seq_len - 1), *score the product of their probabilities; position 0 is skipped, class 1 ("[s]") ends the text. */
int vitx_preprocess_vitstr_u8(const uint8_t *hwc, int nx, int ny, int img_size, float *out_hw);
int vitx_vitstr_decode(const float *probs, int seq_len, int num_classes, int32_t *ids, int *n_ids, double *score);

/* ---- execution context (replaces vit_state + the per-call graph build) ------ */
/* Uploads the weights to `device` in `dtype` and allocates all activation scratch
 * for up to max_batch images once (the reference reallocates per call, vit.cpp:1009-1035).
 * Contexts for >= 16 images cut every batch into 2 contiguous sub-batches that run on two
 * internal HIP streams; the cut is placed where the GEMM tile counts of both parts fill whole
 * rounds of CUs (110 + 146 for 256 ViT-B images on 256 CUs).
 * Results do not depend on the split: images are independent in every kernel.
 * The library reads NO environment variable: everything tunable is in vitx_ctx_options. */
int vitx_ctx_create(const vitx_model *m, int device, int max_batch, int dtype, vitx_ctx **out);
/* Options of a context; every field 0 = the default.  Set struct_size = sizeof(vitx_ctx_options) (lets the struct grow).
 * They change HOW the forward is scheduled or where weights live, not what it computes -- except f16_fast_attention and last_layer_all_rows, which
 * select between two evaluations of the same graph that differ within the operand type's rounding (see the fields). */
typedef struct vitx_ctx_options {
    int32_t struct_size;
    int32_t streams;          /* sub-batch streams, 1..4 (default 2; contexts for fewer than 8 images per stream use 1) */
    int32_t graph;            /* 1: cache the single-stream (small-batch) forward as a hipGraph, captured the second time a call repeats */
    int32_t quant_on_host;    /* 1: expand block-quantised tensors once at upload (16 bits per weight in HBM) instead of keeping the blocks */
    int32_t q4_fused_rows;    /* q4_0 GEMMs with at most this many rows expand the blocks inside the GEMM's LDS-fill path (default 0 = never) */
    int32_t split_first;      /* with 2 streams: images of the first sub-batch (default 0 = the tile-round model decides) */
    int32_t no_ln_fusion;     /* 1: every LayerNorm runs as its own kernel (default: norm2 / the next norm1 ride in the proj / fc2 GEMMs) */
    int32_t ln_test;          /* parity tests only, honoured only as VITX_LN_TEST_KEY | mode (anything else is refused): mode 1 = every fifth tile of a LayerNorm-fusing GEMM behaves as if a peer had timed out, 3 = and withholds its
                                 statistics (real 50 us time-outs): the consumer-side fix-up must then give the same bits; | 4 = the forced fall-backs count against the
                                 fall-back budget (without it a test context keeps fusing whatever the count) */
    int32_t f16_fast_attention; /* VITX_F16 contexts: 1 = q, k, v rounded to fp16 for the attention products (the r03 behaviour: one QKV plane, the fast
                                 attention kernels) instead of the parity mode's f32-grade products (two fp16 planes, three MFMAs per product) */
    int32_t last_layer_all_rows; /* 1 = the last encoder layer computes every token row, as the reference graph does (vit.cpp:805-900 for il = L - 1).
                                 Default 0: past its qkv projection the last layer of a classifier carries only the class-token row of each image -- the only
                                 row vit.cpp:910-911 reads, and no other row can reach it (rows meet only through k and v inside the attention).  The
                                 probabilities are equal within the operand type's rounding (measured on 256 images: 1.1e-3 bf16, 3.0e-4 F16, top-1 equal), not
                                 bit for bit; 0.76 of one layer's work is not done (ViT-B: 6.3 % of the forward's flops).  ViTSTR contexts and contexts with a
                                 residual-stream trace always compute every row. */
} vitx_ctx_options;
#define VITX_LN_TEST_KEY 0x7e570000
int vitx_ctx_create_ex(const vitx_model *m, int device, int max_batch, int dtype, const vitx_ctx_options *options, vitx_ctx **out);
void vitx_ctx_free(vitx_ctx *c);
int vitx_ctx_max_batch(const vitx_ctx *c);
/* Probability rows per image that vitx_forward / vitx_forward_device write: 1 for a classifier ([n][num_classes]), 25 for a ViTSTR
 * file ([n][25][num_classes], row t = token t of the image; decode with vitx_vitstr_decode).  Images are then ONE grey channel:
 * [img_size][img_size] f32, as vitx_preprocess_vitstr_u8 emits. */
int vitx_ctx_out_rows(const vitx_ctx *c);
/* How a forward of n images is cut into contiguous sub-batches (one per internal stream): images[i] = size of sub-batch i, in image
 * order; returns the number of sub-batches (1 when the batch runs on one stream), 0 on a bad argument.  Results never depend on the
 * cut; the parity tests use it to pick the images on either side of every stream boundary.  A batch beyond the kernels' 32-bit buffer window
 * (ViT-B: 3326 images, 2217 in the F16 parity mode) runs as several passes through the same scratch: the cut reported is the first pass's. */
int vitx_ctx_split(const vitx_ctx *c, int n, int32_t *images, int max_parts);

/* Forward pass (replaces vit_encode_image + the compute half of vit_predict,
 * vit.cpp:718-941, 1028-1040) on n <= max_batch images.
 *   imgs_hwc : n x [img_size][img_size][3] f32, as vit_image_preprocess emits
 *   probs    : n x num_classes f32 class probabilities (state.prediction)
 *   logits   : optional (may be NULL), pre-softmax
 * vitx_forward takes host pointers (copies in/out and synchronises);
 * vitx_forward_device takes device pointers and only enqueues on `stream`
 * (a hipStream_t, NULL = the context's own stream).  One exception: the FIRST multi-stream forward of a context (>= 16 images)
 * synchronises the host once for ~0.2 ms while it measures whether its internal sub-batch stream really runs beside the caller's
 * stream (vitx_ctx_stream_retries); every later call, on this or any other caller stream, only enqueues. */
int vitx_forward(vitx_ctx *c, const float *imgs_hwc, int n, float *probs, float *logits);
int vitx_forward_device(vitx_ctx *c, const void *d_imgs_hwc, int n, void *d_probs, void *d_logits, void *stream);
int vitx_ctx_synchronize(vitx_ctx *c);

/* ---- several GPUs in one process (north_star: batch shards + one RCCL gather) -- */
/* One context (replicated weights) and one PERSISTENT host thread per listed device (created here, parked between calls).  Images are
 * cut into contiguous shards, run concurrently, and the results are all-gathered with ONE ncclAllGather over RCCL.
 * The reference has no counterpart (single image, single device: vit.cpp:747).
 *   vitx_group_out_floats     floats per image in `probs`: num_classes, or 25 * num_classes for a ViTSTR file
 *   vitx_group_forward        n host images (f32 HWC, as vit_image_preprocess emits; ViTSTR: one grey plane each); the first n % n_devices
 *                             devices take one extra image; `probs` receives n x out_floats in image order
 *   vitx_group_forward_device the shards are ALREADY on their devices: d_imgs[r] = n_local[r] preprocessed images on devices[r]
 *                             (0 <= n_local[r] <= max_batch_per_device; NULL allowed where n_local[r] == 0).  Nothing crosses PCIe.
 *                             Afterwards EVERY device holds the gathered result of all shards -- vitx_group_result(g, r) is device r's
 *                             copy, vitx_group_result_rows() = n_max = the largest shard:
 *                               topk == 0: [n_devices][n_max][out_floats] f32 probabilities (rows beyond a shard's n_local are zero)
 *                               topk  > 0: [n_devices][n_max][rows_per_image][topk] pairs {f32 probability, i32 class}, descending,
 *                                          ties by the lower class (topk <= 16): 8 k bytes per row on the links instead of 4 num_classes
 *                             Synchronous (returns when every device's stream has finished). */
typedef struct vitx_group vitx_group;
int vitx_group_create(const vitx_model *m, const int *devices, int n_devices, int max_batch_per_device, int dtype, vitx_group **out);
void vitx_group_free(vitx_group *g);
int vitx_group_num_devices(const vitx_group *g);
int vitx_group_out_floats(const vitx_group *g);
int vitx_group_forward(vitx_group *g, const float *imgs_hwc, int n, float *probs);
int vitx_group_forward_device(vitx_group *g, const void *const *d_imgs, const int *n_local, int topk);
const void *vitx_group_result(const vitx_group *g, int device_index);
int vitx_group_result_rows(const vitx_group *g);

/* Sorted top-k of one probability row (vit.cpp:1043-1057: descending by prob). */
int vitx_topk(const float *probs, int num_classes, int k, int32_t *out_idx, float *out_prob);

/* ---- measurement ------------------------------------------------------------ */
/* When enabled, every kernel launch of vitx_forward_device is bracketed by HIP
 * events on the launch stream and the sub-batches run back to back on that one
 * stream (exclusive per-kernel durations); vitx_profile_read() synchronises, folds
 * the event pairs into per-kernel-class totals and clears the pool. */
#define VITX_PROF_MAX_CLASSES 16
typedef struct vitx_prof_entry {
    const char *name;     /* kernel class, e.g. "gemm_fc1_gelu" */
    int32_t launches;
    double total_ms;
    double flops;         /* algorithmic 2*M*N*K summed over the launches (0 for non-GEMM classes) */
    double bytes;         /* algorithmic HBM bytes summed over the launches */
    double busy_ms;       /* wall time during which >= 1 launch of this class was running (union over the
                             context's concurrent sub-batch streams); == total_ms on a single stream */
} vitx_prof_entry;
/* Matrix-pipe probe: back-to-back v_mfma_f32_16x16x32 (the instruction of the GEMM kernels) on register operands on every CU for ~target_ms (no LDS, no memory).
 * fill 0 = zero operands, 1 = constant, 2 = uniform random in [-1, 1).  Returns TFLOP/s and the shader clock the device actually ran
 * at.  MI355X is power-capped on random operands (bf16 ~1830 of the nominal 2517 TFLOP/s): the roofline bench.py reports carries
 * this measured ceiling next to the nominal peak. */
int vitx_probe_mfma(int device, int dtype, int fill, double target_ms, double *tflops, double *clock_mhz);
int vitx_profile_enable(vitx_ctx *c, int on);
int vitx_profile_read(vitx_ctx *c, vitx_prof_entry *out, int max_entries, int *n_entries);
/* What one HIP-event bracket adds to a launch's duration, in microseconds (median of 32 brackets around a 20 us kernel that stamps its own
 * duration, queued back to back on the context's stream): vitx_profile_read() reports RAW event intervals; a caller that wants device-side
 * kernel durations subtracts launches x this (bench.py does, and says so in its line). */
int vitx_profile_bracket_us(vitx_ctx *c, double *bracket_us);

/* ---- single-kernel entry points (device pointers; used by the parity tests) - */
/* y[M][N] (dtype) = LayerNorm(x[M][D] f32) * w + b, eps inside the sqrt (vit.cpp:808-812). */
int vitx_op_layernorm(int dtype, const void *d_x, const void *d_w, const void *d_b, void *d_y, int M, int D, float eps, void *stream);
/* C = A[M][K] . W[N][K]^T with a fused epilogue; A, W in `dtype`.
 *   epi 0: out dtype  = acc + bias                  (vit.cpp:820-821)
 *   epi 1: out dtype  = gelu_tanh(acc + bias)       (vit.cpp:889-893)
 *   epi 2: out f32    = (acc + bias) + out  in place (vit.cpp:868-873, 896-900)
 *   epi 3: out f32    = acc + bias                  (vit.cpp:927-928)
 *   epi 5: out dtype  = TWO planes of acc + bias: hi = round(v) at out[m][n], lo = round((v - hi) * 2048) at out[M * N + m * N + n]
 *                      (the parity mode's f32-grade q, k, v; vitx_op_gemm_ex only; `out` holds 2 * M * N elements)
 * M must be a multiple of 128 rows allocated; N, K multiples of 64. */
int vitx_op_gemm(int dtype, int epi, const void *d_a, const void *d_w, const void *d_bias, void *d_out, int M, int N, int K, void *stream);
/* The same with an explicit kernel family, so every production GEMM variant can be checked against a numeric reference:
 *   kernel 0 = automatic (what the forward would pick for this shape), 1 = ping-pong persistent 256x256 kernel (gemm_pp.hip),
 *   945 / 445 = ring kernels with 256x256 tiles (persistent / one workgroup per tile), 245 = 128x256, 122 = skinny 64x128,
 *   2 = the automatic choice with the tail split forced on (rows of a partial round re-tiled 128x256 in a second launch).
 * Adds epi 4 (patch embedding, vit.cpp:772-797): out f32 [M + M/tpi + 1 rows] : row m -> row m + m/tpi + 1, + d_pos[(m % tpi) + 1][n];
 * d_pos is [tpi + 1][N] f32 and is ignored by the other epilogues.  M_real <= M rows are stored (M is the padded row count).
 * d_w and d_bias must hold N rounded up to a multiple of 256 rows (zeros beyond N).
 * VITX_ERR_UNSUPPORTED when the chosen kernel cannot tile the shape. */
int vitx_op_gemm_ex(int dtype, int epi, int kernel, const void *d_a, const void *d_w, const void *d_bias, void *d_out, const void *d_pos,
                    int M, int M_real, int N, int K, int tpi, void *stream);
/* The residual GEMM with the LayerNorm that follows it computed in its epilogue (what proj + norm2 and fc2 + the next norm1 run as at
 * large batches): d_x f32 [M][N] += A . W^T + bias in place, d_y (dtype) [M][N] = ((x - mean) / sqrt(var + eps)) * ln_w + ln_b of the
 * updated rows.  M % 256 == 0, N in {256, 512, 768, 1024}, K % 128 == 0, at least 128 tiles of 256 x 256 (VITX_ERR_UNSUPPORTED otherwise).
 * test: 0; 1 = every fifth tile behaves as if a peer workgroup had timed out; 3 = and really withholds its statistics (its peers time out
 * after timeout_us microseconds) -- the fix-up launch must then produce the same bits.  Synchronous.  *fallbacks = tiles left to the fix-up. */
int vitx_op_gemm_ln(int dtype, const void *d_a, const void *d_w, const void *d_bias, void *d_x, const void *d_ln_w, const void *d_ln_b, void *d_y,
                    int M, int N, int K, float eps, int test, int timeout_us, int *fallbacks, void *stream);
/* Block-quantised weights on the device (reference: ggml keeps q4_0 ... q8_0 tensors in block form through compute,
 * vit.cpp:384-414, 645-678).  A context built from a quantised file keeps the blocks in HBM (vitx_ctx_weight_bytes reports the
 * footprint; vitx_ctx_options::quant_on_host expands once on the host instead) and expands them on the device:
 *   vitx_op_dequant : out[n_pad][K] (dtype) = expansion of N rows of K/32 blocks of type `qtype` (2 q4_0, 3 q4_1, 6 q5_0, 7 q5_1,
 *                     8 q8_0) laid out as in the file -- except q4_0: d_blocks = nibble plane [N][K/32][16 bytes],
 *                     d_scales = f16 block scales [N][K/32] (d_scales is ignored for the other types); rows N..n_pad are zeros.
 *                     Values are the reference's dequantize_row_* results rounded once (nearest-even) to dtype.
 *   vitx_op_gemm_q4 : C = A[M][K] . dequant(W)^T with the q4_0 blocks expanded in the GEMM's LDS-fill path; d_qs / d_scales as
 *                     above but with N rounded up to 128 rows (zero scales in the pad rows), epi 0..3 as vitx_op_gemm. */
int vitx_op_dequant(int dtype, int qtype, const void *d_blocks, const void *d_scales, void *d_out, int N, int n_pad, int K, void *stream);
int vitx_op_gemm_q4(int dtype, int epi, const void *d_a, const void *d_qs, const void *d_scales, const void *d_bias, void *d_out,
                    int M, int M_real, int N, int K, void *stream);
/* Device bytes held by the context's weight matrices (blocks for quantised tensors, 16-bit operands otherwise). */
size_t vitx_ctx_weight_bytes(const vitx_ctx *c);
/* Contexts of the same loaded model on the same device, with the same operand type and quantisation mode, share ONE device copy of the
 * weights (uploaded by the first, freed with the last: e.g. the two contexts that keep two forwards in flight).
 * 1 when this context attached to a copy that was already there, 0 when it uploaded it. */
int vitx_ctx_shares_weights(const vitx_ctx *c);
/* Diagnostic: GEMM tiles whose fused LayerNorm was left to the fix-up launch since the context was created (a peer workgroup did not
 * publish its row statistics in time -- possible when two such GEMMs on the context's two streams hold each other's CUs; results are
 * the same bits either way).  Synchronises the device.  -1 on error. */
long long vitx_ctx_ln_fallbacks(vitx_ctx *c);
/* Diagnostic: internal sub-batch streams the context re-created because a 40 us probe showed them serialised with the caller's stream
 * (the runtime's stream -> hardware-queue mapping depends on the other streams alive in the process).  0 in a fresh process. */
int vitx_ctx_stream_retries(const vitx_ctx *c);
/* 1 = norm2 / the next norm1 are computed in the proj / fc2 GEMMs' epilogues where the shape allows it (the default on an 8-XCD device); 0 = every
 * LayerNorm is its own launch (option no_ln_fusion, graph cache, a device that does not report 8 XCDs); -1 = the context switched the fusion off
 * itself because more than 8 tiles per forward (averaged over 16 forwards) had to fall back -- peers' CUs held by other work; it tries the fused
 * path again after a cool-down of 256 forwards (doubled on every further trip, at most 65536).  Same bits in all cases. */
int vitx_ctx_ln_fusion_active(const vitx_ctx *c);
/* out[n_img*N][D] (dtype) = softmax(q k^T / sqrt(64)) v per head from qkv[n_img*N][3D] (vit.cpp:826-866). */
int vitx_op_attention(int dtype, const void *d_qkv, void *d_out, int n_img, int N, int D, int H, void *stream);
/* kernel 0 = automatic, 1 = single-pass kernel (N <= 224, 257-288 or 577-608 tokens only),
 * 3 = pipelined two-pass kernel (any N; LDS-DMA double buffering, transposed LDS reads), 4 = persistent single-pass kernel (193..224
 * tokens: one workgroup per CU walks the (image, head) items, the next item's K/V land by LDS-DMA while the current one is computed).
 * F16: kernels 1 and 3 give bit-identical results (both apply the fp16 exp table relative to the TRUE row maximum).  BF16: they do not --
 * kernel 3 keeps a RUNNING maximum and rounds its numerators at another scale (equal within the numerators' bf16 rounding, 4e-3 relative in
 * the tests), and the automatic choice switches from kernel 1 / 4 to kernel 3 above 288 tokens: a bf16 result depends on the kernel family
 * the token count selects, not on the batch.  Kernel 4 issues v_mfma_f32_16x16x32 instead of 32x32x16 (same products, another accumulation
 * grouping: equal within f32 summation noise).
 * kernel 5 = streaming two-pass kernel (attention_stream.hip; any N, head dim 64: v_mfma_f32_16x16x32, 64-key chunks through a 3-slot LDS-DMA ring). */
int vitx_op_attention_ex(int dtype, int kernel, const void *d_qkv, void *d_out, int n_img, int N, int D, int H, void *stream);
/* Token 0 of every image only -- the one attention row the last layer of a classifier needs (vit.cpp:910-911): d_out [n_img][D] (dtype),
 * softmax(q_0 k^T / sqrt(head_dim)) v per head with f32 products and the same numerator rounding as the kernels above.  lo_off != 0: d_qkv is
 * the hi plane of the F16 parity mode and the lo plane lies lo_off elements behind it (multiple of 8; VITX_F16 only).  head_dim 8, 16, 32, 64, 128. */
int vitx_op_attention_cls(int dtype, const void *d_qkv, long lo_off, void *d_out, int n_img, int N, int D, int H, void *stream);
/* The F16 parity mode's attention on f32 q, k, v (the reference multiplies f32 operands, vit.cpp:848,858): d_qkv_f32 [n_img * N][3 D] f32 is
 * split into hi / lo fp16 planes (what the QKV GEMM's epi 5 emits) and every product is hi.hi + (hi.lo + lo.hi) / 2048.  d_out [n_img * N][D]
 * fp16.  Head dim 64.  TEST-ONLY entry point: it allocates and frees its own scratch and synchronises `stream` on every call. */
int vitx_op_attention_f32(const float *d_qkv_f32, void *d_out, int n_img, int N, int D, int H, void *stream);
/* The same kernel on planes that are already split (the output of vitx_op_gemm_ex epi 5): d_hi [n_img * N][3 D] fp16, the lo plane lo_off
 * ELEMENTS behind it (a multiple of 4, at least n_img * N * 3 D, both planes below 0xf0000000 bytes).  Only enqueues on `stream`. */
int vitx_op_attention_planes(const void *d_hi, long lo_off, void *d_out, int n_img, int N, int D, int H, void *stream);
/* probs = softmax(logits) over num_classes with the reference's fp16 exp rounding (vit.cpp:931). */
int vitx_op_softmax(const void *d_logits, void *d_probs, int rows, int cols, int ld, void *stream);
/* The same with the rounding type of the exp explicit (VITX_F16 = the reference's LUT semantics, VITX_BF16 = the bf16 engine). */
int vitx_op_softmax_dt(int dtype, const void *d_logits, void *d_probs, int rows, int cols, int ld, void *stream);

/* ---- residual-stream trace (parity localisation) ------------------------------ */
/* After vitx_trace_enable(ctx, ids, n) every forward also copies the f32 residual stream X of images ids[0..n) -- after the
 * patch embedding and after each encoder layer -- into a device buffer; vitx_trace_read() synchronises and returns it as
 * [L + 1][n][tokens][hidden] f32 (vit.cpp:797 and :900: the tensor `cur` carries between blocks).  n = 0 disables.
 * A traced context evaluates EVERY row of the last layer (as with last_layer_all_rows = 1): its probabilities are those of the whole graph and
 * can differ from the same context's untraced forward by the operand type's rounding (bf16 about 1e-3). */
int vitx_trace_enable(vitx_ctx *c, const int32_t *image_ids, int n);
int vitx_trace_read(vitx_ctx *c, float *out, size_t n_floats);

#ifdef __cplusplus
}
#endif
#endif /* VITX_H */

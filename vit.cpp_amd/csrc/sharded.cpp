// sharded.cpp -- one process, several GPUs: the C / C++ multi-GPU entry point of the forward path.
//
// The reference has no multi-device code (batch is hard-wired to 1, vit.cpp:747); north_star adds data parallelism:
// images are independent (no cross-image op in vit_encode_image, vit.cpp:718-941), so a batch is cut into contiguous
// shards, one per GPU, weights are replicated, and the ONLY collective is one all-gather of the class probabilities
// ([n_local, num_classes] f32 per GPU; 1 MB at 256 images, latency-bound on xGMI).  This file is that schedule for a C++
// caller: one vitx_ctx + one host thread per GPU, RCCL (ncclAllGather) called directly -- the Python route
// (vit.cpp_amd/dist.py, one process per GPU over torch.distributed) does the same thing for bench.py.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <string.h>

#include <memory>
#include <new>
#include <thread>
#include <vector>

#include "model_file.h"

using namespace vitx;

struct vitx_group {
    std::vector<int> devices;
    std::vector<vitx_ctx *> ctx;
    std::vector<ncclComm_t> comm;
    std::vector<hipStream_t> stream;
    std::vector<float *> d_img, d_probs, d_all;     // per device: shard images, shard probabilities (padded to n_max), gathered [ndev][n_max][C]
    int max_per_dev = 0, C = 0, S = 0, Cin = 3;      // C = probability floats per image
    ~vitx_group() {
        for (size_t i = 0; i < devices.size(); ++i) {
            (void)hipSetDevice(devices[i]);
            if (i < comm.size() && comm[i]) (void)ncclCommDestroy(comm[i]);
            if (i < d_img.size() && d_img[i]) (void)hipFree(d_img[i]);
            if (i < d_probs.size() && d_probs[i]) (void)hipFree(d_probs[i]);
            if (i < d_all.size() && d_all[i]) (void)hipFree(d_all[i]);
            if (i < stream.size() && stream[i]) (void)hipStreamDestroy(stream[i]);
            if (i < ctx.size() && ctx[i]) vitx_ctx_free(ctx[i]);
        }
    }
};

extern "C" {

int vitx_group_create(const vitx_model *m, const int *devices, int n_devices, int max_batch_per_device, int dtype, vitx_group **out) {
    if (!m || !devices || !out || n_devices <= 0 || max_batch_per_device <= 0) { set_error("vitx_group_create: invalid argument"); return VITX_ERR_ARG; }
    *out = nullptr;
    for (int i = 0; i < n_devices; ++i)
        for (int j = 0; j < i; ++j)
            if (devices[i] == devices[j]) { set_error("vitx_group_create: device %d listed twice", devices[i]); return VITX_ERR_ARG; }
    std::unique_ptr<vitx_group> g(new (std::nothrow) vitx_group());
    if (!g) return VITX_ERR_NOMEM;
    g->devices.assign(devices, devices + n_devices);
    g->ctx.assign(n_devices, nullptr); g->comm.assign(n_devices, nullptr); g->stream.assign(n_devices, nullptr);
    g->d_img.assign(n_devices, nullptr); g->d_probs.assign(n_devices, nullptr); g->d_all.assign(n_devices, nullptr);
    // per image: in_chans planes in, out_rows x num_classes probabilities out (a ViTSTR file: 1 grey plane, 25 rows)
    g->max_per_dev = max_batch_per_device; g->C = m->hp.num_classes * (vitx_model_seq_len(m) ? vitx_model_seq_len(m) : 1); g->S = m->hp.img_size; g->Cin = m->in_chans;
    const size_t img_floats = (size_t)g->S * g->S * g->Cin;
    for (int i = 0; i < n_devices; ++i) {
        int rc = vitx_ctx_create(m, devices[i], max_batch_per_device, dtype, &g->ctx[i]);      // replicated weights, one context per GPU
        if (rc != VITX_OK) return rc;
        if (hipSetDevice(devices[i]) != hipSuccess || hipStreamCreateWithFlags(&g->stream[i], hipStreamNonBlocking) != hipSuccess ||
            hipMalloc((void **)&g->d_img[i], (size_t)max_batch_per_device * img_floats * 4) != hipSuccess ||
            hipMalloc((void **)&g->d_probs[i], (size_t)max_batch_per_device * g->C * 4) != hipSuccess ||
            hipMalloc((void **)&g->d_all[i], (size_t)n_devices * max_batch_per_device * g->C * 4) != hipSuccess) {
            set_error("vitx_group_create: device %d: %s", devices[i], hipGetErrorString(hipGetLastError())); return VITX_ERR_HIP;
        }
        if (hipMemset(g->d_probs[i], 0, (size_t)max_batch_per_device * g->C * 4) != hipSuccess) return VITX_ERR_HIP;
    }
    const ncclResult_t nr = ncclCommInitAll(g->comm.data(), n_devices, g->devices.data());
    if (nr != ncclSuccess) { set_error("vitx_group_create: ncclCommInitAll: %s", ncclGetErrorString(nr)); return VITX_ERR_HIP; }
    *out = g.release();
    return VITX_OK;
}

void vitx_group_free(vitx_group *g) { delete g; }
int vitx_group_num_devices(const vitx_group *g) { return g ? (int)g->devices.size() : 0; }

// contiguous shard [lo, hi) of device r: the first n % ndev devices take one extra image (same rule as dist.shard_bounds)
static void shard(int n, int ndev, int r, int *lo, int *hi) {
    const int base = n / ndev, extra = n % ndev;
    *lo = r * base + (r < extra ? r : extra);
    *hi = *lo + base + (r < extra ? 1 : 0);
}

int vitx_group_forward(vitx_group *g, const float *imgs_hwc, int n, float *probs) {
    if (!g || !imgs_hwc || !probs || n <= 0) { set_error("vitx_group_forward: invalid argument"); return VITX_ERR_ARG; }
    const int ndev = (int)g->devices.size();
    const int n_max = (n + ndev - 1) / ndev;
    if (n_max > g->max_per_dev) { set_error("vitx_group_forward: %d images over %d GPUs exceeds %d per GPU", n, ndev, g->max_per_dev); return VITX_ERR_ARG; }
    const size_t img_floats = (size_t)g->S * g->S * g->Cin;
    const int C = g->C;
    std::vector<int> rc(ndev, VITX_OK);
    std::vector<std::string> err(ndev);
    auto worker = [&](int r) {          // one host thread per GPU: H2D of its shard, forward, then the one collective, all on its stream
        int lo, hi; shard(n, ndev, r, &lo, &hi);
        const int nl = hi - lo;
        auto fail = [&](int code, const char *what, const char *detail) { rc[r] = code; err[r] = std::string(what) + ": " + detail; };
        if (hipSetDevice(g->devices[r]) != hipSuccess) { fail(VITX_ERR_HIP, "hipSetDevice", hipGetErrorString(hipGetLastError())); }
        hipStream_t st = g->stream[r];
        if (rc[r] == VITX_OK && nl > 0) {
            if (hipMemcpyAsync(g->d_img[r], imgs_hwc + (size_t)lo * img_floats, (size_t)nl * img_floats * 4, hipMemcpyHostToDevice, st) != hipSuccess) fail(VITX_ERR_HIP, "H2D", hipGetErrorString(hipGetLastError()));
            else if (int e = vitx_forward_device(g->ctx[r], g->d_img[r], nl, g->d_probs[r], nullptr, st)) fail(e, "vitx_forward_device", vitx_last_error());
        }
        // EVERY rank joins the collective, even after a local failure (its shard is then garbage and the call reports the error):
        // a rank that skipped it would hang the others
        const ncclResult_t nr = ncclAllGather(g->d_probs[r], g->d_all[r], (size_t)n_max * C, ncclFloat, g->comm[r], st);
        if (nr != ncclSuccess && rc[r] == VITX_OK) fail(VITX_ERR_HIP, "ncclAllGather", ncclGetErrorString(nr));
        if (hipStreamSynchronize(st) != hipSuccess && rc[r] == VITX_OK) fail(VITX_ERR_HIP, "hipStreamSynchronize", hipGetErrorString(hipGetLastError()));
    };
    std::vector<std::thread> th;
    for (int r = 1; r < ndev; ++r) th.emplace_back(worker, r);
    worker(0);
    for (auto &t : th) t.join();
    for (int r = 0; r < ndev; ++r) if (rc[r] != VITX_OK) { set_error("vitx_group_forward: device %d: %s", g->devices[r], err[r].c_str()); return rc[r]; }
    // every GPU now holds all shards ([ndev][n_max][C], ragged shards zero-padded); device 0's copy goes back in image order
    if (hipSetDevice(g->devices[0]) != hipSuccess) return VITX_ERR_HIP;
    for (int r = 0; r < ndev; ++r) {
        int lo, hi; shard(n, ndev, r, &lo, &hi);
        if (hi > lo && hipMemcpy(probs + (size_t)lo * C, g->d_all[0] + (size_t)r * n_max * C, (size_t)(hi - lo) * C * 4, hipMemcpyDeviceToHost) != hipSuccess) {
            set_error("vitx_group_forward: D2H: %s", hipGetErrorString(hipGetLastError())); return VITX_ERR_HIP;
        }
    }
    return VITX_OK;
}

}  // extern "C"

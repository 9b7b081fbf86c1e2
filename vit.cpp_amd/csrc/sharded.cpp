// sharded.cpp -- one process, several GPUs: the C / C++ multi-GPU entry point of the forward path.
//
// The reference has no multi-device code (batch is hard-wired to 1, vit.cpp:747); north_star adds data parallelism:
// images are independent (no cross-image op in vit_encode_image, vit.cpp:718-941), so a batch is cut into contiguous
// shards, one per GPU, weights are replicated, and the ONLY collective is one all-gather of the results -- the class
// probabilities ([n_local, num_classes] f32 per GPU; 1 MB at 256 images) or, on request, the device-side top-k
// ([n_local, k] {f32, i32} pairs: 10 KB at k = 5), latency-bound on xGMI either way.  This file is that schedule for a C++
// caller: one vitx_ctx + one PERSISTENT host thread per GPU (created with the group, parked on a condition variable between
// calls), RCCL (ncclAllGather) called directly -- the Python route (vit.cpp_amd/dist.py, one process per GPU over
// torch.distributed) does the same thing for bench.py.  Shards may already be resident on their devices
// (vitx_group_forward_device): nothing then crosses PCIe but the caller's final read of the gathered result.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "kernels.h"
#include "model_file.h"

using namespace vitx;

struct vitx_group {
    std::vector<int> devices;
    std::vector<vitx_ctx *> ctx;
    std::vector<ncclComm_t> comm;
    std::vector<hipStream_t> stream;
    std::vector<float *> d_img, d_probs;            // per device: staging for host-fed shards; shard probabilities [max_per_dev][C]
    std::vector<void *> d_send, d_all;              // per device: this shard's gather payload (padded to n_max rows); gathered [ndev][n_max][payload]
    int max_per_dev = 0, C = 0, S = 0, Cin = 3;      // C = probability floats per image (num_classes, x 25 rows for a ViTSTR file)
    int rows_per_img = 1, classes = 0;
    int n_max = 0, topk = 0;                         // of the last call
    // ---- persistent workers: job = (sequence number, per-device arguments); a worker runs every job exactly once
    struct Job { const void *const *d_imgs = nullptr; const float *h_imgs = nullptr; const int *n_local = nullptr; const int *lo = nullptr; int n_max = 0, topk = 0; };
    Job job;
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    unsigned long long seq = 0;
    int pending = 0;
    bool quit = false;
    std::vector<std::thread> workers;
    std::vector<int> rc;
    std::vector<std::string> err;

    size_t payload_floats(int topk_) const { return topk_ > 0 ? (size_t)rows_per_img * topk_ * 2 : (size_t)C; }      // 4-byte words per image in the gather

    void run(int r) {          // one shard: (H2D of host-fed images,) forward, (top-k,) then the one collective, all on this device's stream
        const Job j = job;
        const int nl = j.n_local[r];
        auto fail = [&](int code, const char *what, const char *detail) { if (rc[r] == VITX_OK) { rc[r] = code; err[r] = std::string(what) + ": " + detail; } };
        if (hipSetDevice(devices[r]) != hipSuccess) fail(VITX_ERR_HIP, "hipSetDevice", hipGetErrorString(hipGetLastError()));
        hipStream_t st = stream[r];
        const size_t img_floats = (size_t)S * S * Cin, pw = payload_floats(j.topk);
        if (rc[r] == VITX_OK && nl > 0) {
            const void *src = j.d_imgs ? j.d_imgs[r] : d_img[r];
            if (!j.d_imgs && hipMemcpyAsync(d_img[r], j.h_imgs + (size_t)j.lo[r] * img_floats, (size_t)nl * img_floats * 4, hipMemcpyHostToDevice, st) != hipSuccess)
                fail(VITX_ERR_HIP, "H2D", hipGetErrorString(hipGetLastError()));
            else if (!src) fail(VITX_ERR_ARG, "shard", "NULL device pointer for a non-empty shard");
            else if (int e = vitx_forward_device(ctx[r], src, nl, d_probs[r], nullptr, st)) fail(e, "vitx_forward_device", vitx_last_error());
            else if (j.topk > 0) {
                const hipError_t he = launch_topk(d_probs[r], nl * rows_per_img, classes, j.topk, d_send[r], st);
                if (he != hipSuccess) fail(VITX_ERR_HIP, "launch_topk", hipGetErrorString(he));
            }
        }
        // the payload of a ragged or empty shard is zero-padded to n_max rows so that every rank contributes the same count
        const void *send = j.topk > 0 ? d_send[r] : (const void *)d_probs[r];
        if (nl < j.n_max && hipMemsetAsync((char *)const_cast<void *>(send) + (size_t)nl * pw * 4, 0, (size_t)(j.n_max - nl) * pw * 4, st) != hipSuccess)
            fail(VITX_ERR_HIP, "hipMemsetAsync", hipGetErrorString(hipGetLastError()));
        // EVERY rank joins the collective, even after a local failure (its shard is then garbage and the call reports the error):
        // a rank that skipped it would hang the others
        const ncclResult_t nr = ncclAllGather(send, d_all[r], (size_t)j.n_max * pw, ncclFloat, comm[r], st);
        if (nr != ncclSuccess) fail(VITX_ERR_HIP, "ncclAllGather", ncclGetErrorString(nr));
        if (hipStreamSynchronize(st) != hipSuccess) fail(VITX_ERR_HIP, "hipStreamSynchronize", hipGetErrorString(hipGetLastError()));
    }
    void worker(int r) {
        unsigned long long seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_go.wait(lk, [&] { return quit || seq != seen; });
                if (quit) return;
                seen = seq;
            }
            run(r);
            {
                std::lock_guard<std::mutex> lk(mu);
                if (--pending == 0) cv_done.notify_one();
            }
        }
    }
    int dispatch(const Job &j) {          // post the job, run shard 0 on the calling thread, wait for the others
        const int ndev = (int)devices.size();
        for (int r = 0; r < ndev; ++r) { rc[r] = VITX_OK; err[r].clear(); }
        {
            std::lock_guard<std::mutex> lk(mu);
            job = j; pending = ndev - 1; ++seq;
        }
        cv_go.notify_all();
        run(0);
        {
            std::unique_lock<std::mutex> lk(mu);
            cv_done.wait(lk, [&] { return pending == 0; });
        }
        n_max = j.n_max; topk = j.topk;
        for (int r = 0; r < ndev; ++r) if (rc[r] != VITX_OK) { set_error("vitx_group_forward: device %d: %s", devices[r], err[r].c_str()); return rc[r]; }
        return VITX_OK;
    }
    ~vitx_group() {
        {
            std::lock_guard<std::mutex> lk(mu);
            quit = true;
        }
        cv_go.notify_all();
        for (auto &t : workers) if (t.joinable()) t.join();
        for (size_t i = 0; i < devices.size(); ++i) {
            (void)hipSetDevice(devices[i]);
            if (i < comm.size() && comm[i]) (void)ncclCommDestroy(comm[i]);
            if (i < d_img.size() && d_img[i]) (void)hipFree(d_img[i]);
            if (i < d_probs.size() && d_probs[i]) (void)hipFree(d_probs[i]);
            if (i < d_send.size() && d_send[i]) (void)hipFree(d_send[i]);
            if (i < d_all.size() && d_all[i]) (void)hipFree(d_all[i]);
            if (i < stream.size() && stream[i]) (void)hipStreamDestroy(stream[i]);
            if (i < ctx.size() && ctx[i]) vitx_ctx_free(ctx[i]);
        }
    }
};

extern "C" {

#define VITX_GROUP_MAX_TOPK 16

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
    g->d_img.assign(n_devices, nullptr); g->d_probs.assign(n_devices, nullptr); g->d_send.assign(n_devices, nullptr); g->d_all.assign(n_devices, nullptr);
    g->rc.assign(n_devices, VITX_OK); g->err.assign(n_devices, std::string());
    // per image: in_chans planes in, out_rows x num_classes probabilities out (a ViTSTR file: 1 grey plane, 25 rows)
    g->rows_per_img = vitx_model_seq_len(m) ? vitx_model_seq_len(m) : 1; g->classes = m->hp.num_classes;
    g->max_per_dev = max_batch_per_device; g->C = g->classes * g->rows_per_img; g->S = m->hp.img_size; g->Cin = m->in_chans;
    const size_t img_floats = (size_t)g->S * g->S * g->Cin;
    const size_t send_words = std::max((size_t)g->C, (size_t)g->rows_per_img * VITX_GROUP_MAX_TOPK * 2);
    for (int i = 0; i < n_devices; ++i) {
        int rc = vitx_ctx_create(m, devices[i], max_batch_per_device, dtype, &g->ctx[i]);      // replicated weights, one context per GPU
        if (rc != VITX_OK) return rc;
        if (hipSetDevice(devices[i]) != hipSuccess || hipStreamCreateWithFlags(&g->stream[i], hipStreamNonBlocking) != hipSuccess ||
            hipMalloc((void **)&g->d_img[i], (size_t)max_batch_per_device * img_floats * 4) != hipSuccess ||
            hipMalloc((void **)&g->d_probs[i], (size_t)max_batch_per_device * g->C * 4) != hipSuccess ||
            hipMalloc(&g->d_send[i], (size_t)max_batch_per_device * g->rows_per_img * VITX_GROUP_MAX_TOPK * 8) != hipSuccess ||
            hipMalloc(&g->d_all[i], (size_t)n_devices * max_batch_per_device * send_words * 4) != hipSuccess) {
            set_error("vitx_group_create: device %d: %s", devices[i], hipGetErrorString(hipGetLastError())); return VITX_ERR_HIP;
        }
        if (hipMemset(g->d_probs[i], 0, (size_t)max_batch_per_device * g->C * 4) != hipSuccess) return VITX_ERR_HIP;
    }
    const ncclResult_t nr = ncclCommInitAll(g->comm.data(), n_devices, g->devices.data());
    if (nr != ncclSuccess) { set_error("vitx_group_create: ncclCommInitAll: %s", ncclGetErrorString(nr)); return VITX_ERR_HIP; }
    vitx_group *gp = g.get();
    for (int r = 1; r < n_devices; ++r) g->workers.emplace_back([gp, r] { gp->worker(r); });       // device 0's shard runs on the calling thread
    *out = g.release();
    return VITX_OK;
}

void vitx_group_free(vitx_group *g) { delete g; }
int vitx_group_num_devices(const vitx_group *g) { return g ? (int)g->devices.size() : 0; }
int vitx_group_out_floats(const vitx_group *g) { return g ? g->C : 0; }

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
    std::vector<int> nl(ndev), lo(ndev);
    for (int r = 0; r < ndev; ++r) { int a, b; shard(n, ndev, r, &a, &b); lo[r] = a; nl[r] = b - a; }
    vitx_group::Job j; j.h_imgs = imgs_hwc; j.n_local = nl.data(); j.lo = lo.data(); j.n_max = n_max; j.topk = 0;
    const int rc = g->dispatch(j);
    if (rc != VITX_OK) return rc;
    // every GPU now holds all shards ([ndev][n_max][C], ragged shards zero-padded); device 0's copy goes back in image order
    if (hipSetDevice(g->devices[0]) != hipSuccess) return VITX_ERR_HIP;
    const int C = g->C;
    for (int r = 0; r < ndev; ++r) {
        if (nl[r] > 0 && hipMemcpy(probs + (size_t)lo[r] * C, (const float *)g->d_all[0] + (size_t)r * n_max * C, (size_t)nl[r] * C * 4, hipMemcpyDeviceToHost) != hipSuccess) {
            set_error("vitx_group_forward: D2H: %s", hipGetErrorString(hipGetLastError())); return VITX_ERR_HIP;
        }
    }
    return VITX_OK;
}

int vitx_group_forward_device(vitx_group *g, const void *const *d_imgs, const int *n_local, int topk) {
    if (!g || !d_imgs || !n_local || topk < 0 || topk > VITX_GROUP_MAX_TOPK || topk > (g ? g->classes : 0)) { set_error("vitx_group_forward_device: invalid argument (top-k 0..%d)", VITX_GROUP_MAX_TOPK); return VITX_ERR_ARG; }
    const int ndev = (int)g->devices.size();
    int n_max = 0;
    for (int r = 0; r < ndev; ++r) {
        if (n_local[r] < 0 || n_local[r] > g->max_per_dev) { set_error("vitx_group_forward_device: shard %d holds %d images, capacity %d", r, n_local[r], g->max_per_dev); return VITX_ERR_ARG; }
        if (n_local[r] > 0 && !d_imgs[r]) { set_error("vitx_group_forward_device: shard %d: NULL device pointer", r); return VITX_ERR_ARG; }
        n_max = std::max(n_max, n_local[r]);
    }
    if (n_max == 0) { set_error("vitx_group_forward_device: no images"); return VITX_ERR_ARG; }
    vitx_group::Job j; j.d_imgs = d_imgs; j.n_local = n_local; j.n_max = n_max; j.topk = topk;
    return g->dispatch(j);
}

const void *vitx_group_result(const vitx_group *g, int device_index) { return (g && device_index >= 0 && device_index < (int)g->devices.size()) ? g->d_all[device_index] : nullptr; }
int vitx_group_result_rows(const vitx_group *g) { return g ? g->n_max : 0; }

}  // extern "C"

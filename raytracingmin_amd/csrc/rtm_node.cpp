// rtm_node.cpp — one-process, N-GPU rendering for the host program: the image is cut into row
// strips on 8-row tile boundaries, every GPU of the node renders its strip through the C ABI
// (rtm_render_device, one host thread per device), and the strips are collected on device 0 with ONE
// grouped RCCL exchange over xGMI (ncclSend from every other device, matching ncclRecv on device 0).
// The reference has no multi-device code; this is the north_star's "image tiled across the 8 GPUs of
// one node with a single RCCL gather".  Linked into rtm_cli only — librtm_hip.so itself stays free of
// RCCL so that it can share a process with PyTorch's bundled copy.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rtm.h"
#include "rtm_node.h"

namespace {
struct Strip {
    int begin, end;
};
// same rule as raytracingmin_amd/distributed.py::partition_rows
std::vector<Strip> partition_rows(int height, int parts) {
    const int tiles = (height + 7) / 8;
    std::vector<Strip> out;
    for (int r = 0; r < parts; ++r) {
        const long t0 = (long)r * tiles / parts, t1 = (long)(r + 1) * tiles / parts;
        Strip s{(int)(t0 * 8 < height ? t0 * 8 : height), (int)(t1 * 8 < height ? t1 * 8 : height)};
        out.push_back(s);
    }
    return out;
}
#define NODE_HIP(expr)                                                                      \
    do {                                                                                    \
        hipError_t e__ = (expr);                                                            \
        if (e__ != hipSuccess) {                                                            \
            err = std::string(#expr) + ": " + hipGetErrorString(e__);                       \
            return RTM_ERR_HIP;                                                             \
        }                                                                                   \
    } while (0)
#define NODE_NCCL(expr)                                                                     \
    do {                                                                                    \
        ncclResult_t r__ = (expr);                                                          \
        if (r__ != ncclSuccess) {                                                           \
            err = std::string(#expr) + ": " + ncclGetErrorString(r__);                      \
            return RTM_ERR_HIP;                                                             \
        }                                                                                   \
    } while (0)
}  // namespace

int rtm_node_render_u8(const rtm_settings* st, const rtm_sphere* spheres, size_t n, const rtm_options* base,
                       int n_devices, int virtual_strips, uint8_t* out_u8_host, rtm_stats* total,
                       std::string& err) {
    if (!st || !base || !out_u8_host || n_devices < 1) return RTM_ERR_INVALID_ARGUMENT;
    const int W = st->width, H = st->height;
    // virtual_strips > 0: that many strips, all on base->device (exercises the tiling on one GPU)
    const int parts = virtual_strips > 0 ? virtual_strips : n_devices;
    const std::vector<Strip> strips = partition_rows(H, parts);
    std::vector<int> dev(parts);
    for (int r = 0; r < parts; ++r) dev[r] = virtual_strips > 0 ? base->device : r;
    const int root = dev[0];

    std::vector<uint8_t*> d_strip(parts, nullptr);
    std::vector<hipStream_t> stream(parts, nullptr);
    std::vector<rtm_stats> stats(parts);
    std::vector<int> rc(parts, RTM_OK);
    std::vector<std::string> detail(parts);
    for (int r = 0; r < parts; ++r) {
        NODE_HIP(hipSetDevice(dev[r]));
        NODE_HIP(hipStreamCreate(&stream[r]));
        const size_t bytes = (size_t)(strips[r].end - strips[r].begin) * W * 3;
        if (bytes) NODE_HIP(hipMalloc((void**)&d_strip[r], bytes));
    }
    // one host thread per strip: the renders of different GPUs run concurrently
    auto work = [&](int r) {
        rtm_options opt = *base;
        opt.device = dev[r];
        opt.row_begin = strips[r].begin;
        opt.row_end = strips[r].end;
        rc[r] = rtm_render_device(st, spheres, n, 0, &opt, nullptr, nullptr, d_strip[r], stream[r], &stats[r]);
        if (rc[r] != RTM_OK) detail[r] = rtm_last_error_detail();
    };
    if (virtual_strips > 0) {
        for (int r = 0; r < parts; ++r) work(r);
    } else {
        std::vector<std::thread> th;
        for (int r = 0; r < parts; ++r) th.emplace_back(work, r);
        for (auto& t : th) t.join();
    }
    for (int r = 0; r < parts; ++r)
        if (rc[r] != RTM_OK) {
            err = "strip " + std::to_string(r) + ": " + detail[r];
            return rc[r];
        }

    // gather on the root device
    uint8_t* d_full = nullptr;
    NODE_HIP(hipSetDevice(root));
    NODE_HIP(hipMalloc((void**)&d_full, (size_t)W * H * 3));
    if (virtual_strips > 0 || n_devices == 1) {
        for (int r = 0; r < parts; ++r) {
            const size_t bytes = (size_t)(strips[r].end - strips[r].begin) * W * 3;
            if (bytes)
                NODE_HIP(hipMemcpyAsync(d_full + (size_t)strips[r].begin * W * 3, d_strip[r], bytes,
                                        hipMemcpyDeviceToDevice, stream[0]));
        }
        NODE_HIP(hipStreamSynchronize(stream[0]));
    } else {
        std::vector<ncclComm_t> comm(parts);
        NODE_NCCL(ncclCommInitAll(comm.data(), parts, dev.data()));
        NODE_NCCL(ncclGroupStart());
        for (int r = 0; r < parts; ++r) {
            const size_t bytes = (size_t)(strips[r].end - strips[r].begin) * W * 3;
            if (!bytes) continue;
            NODE_NCCL(ncclSend(d_strip[r], bytes, ncclUint8, 0, comm[r], stream[r]));
            NODE_NCCL(ncclRecv(d_full + (size_t)strips[r].begin * W * 3, bytes, ncclUint8, r, comm[0], stream[0]));
        }
        NODE_NCCL(ncclGroupEnd());
        for (int r = 0; r < parts; ++r) {
            NODE_HIP(hipSetDevice(dev[r]));
            NODE_HIP(hipStreamSynchronize(stream[r]));
        }
        for (auto c : comm) ncclCommDestroy(c);
    }
    NODE_HIP(hipSetDevice(root));
    NODE_HIP(hipMemcpy(out_u8_host, d_full, (size_t)W * H * 3, hipMemcpyDeviceToHost));
    (void)hipFree(d_full);
    if (total) std::memset(total, 0, sizeof *total);
    for (int r = 0; r < parts; ++r) {
        (void)hipSetDevice(dev[r]);
        if (d_strip[r]) (void)hipFree(d_strip[r]);
        (void)hipStreamDestroy(stream[r]);
        if (total) {
            total->samples += stats[r].samples;
            total->casts += stats[r].casts;
            total->bounces += stats[r].bounces;
            total->draws += stats[r].draws;
            // strips on different GPUs overlap in time: the frame's kernel time is the longest strip;
            // virtual strips on one GPU run back to back
            total->kernel_ms = virtual_strips > 0 ? total->kernel_ms + stats[r].kernel_ms
                                                  : (stats[r].kernel_ms > total->kernel_ms ? stats[r].kernel_ms : total->kernel_ms);
        }
    }
    return RTM_OK;
}

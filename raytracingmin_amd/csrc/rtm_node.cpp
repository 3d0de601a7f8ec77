// rtm_node.cpp — one-process, N-GPU rendering for the host program: the image is dealt out in
// interleaved 8-row bands (band b goes to GPU b mod N: every GPU gets the same mix of cheap and costly
// rows), every GPU of the node renders its bands through the C ABI (rtm_render_device with
// band_count/band_index, one host thread per device), and the band stacks are collected on device 0
// with ONE grouped RCCL exchange over xGMI (ncclSend from every other device, matching ncclRecv on
// device 0), where the bands are put back in image order.
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
    // virtual_strips > 0: that many parts, all on base->device (exercises the tiling on one GPU)
    const int parts = virtual_strips > 0 ? virtual_strips : n_devices;
    std::vector<rtm_options> opts(parts, *base);
    std::vector<size_t> part_bytes(parts), stage_off(parts + 1, 0);
    std::vector<int> dev(parts);
    for (int r = 0; r < parts; ++r) {
        dev[r] = virtual_strips > 0 ? base->device : r;
        opts[r].device = dev[r];
        opts[r].row_begin = 0;
        opts[r].row_end = H;
        opts[r].band_count = parts;
        opts[r].band_index = r;
        part_bytes[r] = (size_t)rtm_output_rows(&opts[r]) * W * 3;
        stage_off[r + 1] = stage_off[r] + part_bytes[r];
    }
    const int root = dev[0];

    std::vector<uint8_t*> d_strip(parts, nullptr);
    std::vector<hipStream_t> stream(parts, nullptr);
    std::vector<rtm_stats> stats(parts);
    std::vector<int> rc(parts, RTM_OK);
    std::vector<std::string> detail(parts);
    for (int r = 0; r < parts; ++r) {
        NODE_HIP(hipSetDevice(dev[r]));
        NODE_HIP(hipStreamCreate(&stream[r]));
        if (part_bytes[r]) NODE_HIP(hipMalloc((void**)&d_strip[r], part_bytes[r]));
    }
    // one host thread per strip: the renders of different GPUs run concurrently
    auto work = [&](int r) {
        rc[r] = rtm_render_device(st, spheres, n, 0, &opts[r], nullptr, nullptr, d_strip[r], stream[r], &stats[r]);
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
            err = "part " + std::to_string(r) + ": " + detail[r];
            return rc[r];
        }

    // gather on the root device: the band stacks land side by side in a staging buffer ...
    uint8_t *d_full = nullptr, *d_stage = nullptr;
    NODE_HIP(hipSetDevice(root));
    NODE_HIP(hipMalloc((void**)&d_full, (size_t)W * H * 3));
    std::vector<const uint8_t*> src(parts);
    if (virtual_strips > 0 || n_devices == 1) {
        for (int r = 0; r < parts; ++r) src[r] = d_strip[r];
    } else {
        NODE_HIP(hipMalloc((void**)&d_stage, stage_off[parts]));
        std::vector<ncclComm_t> comm(parts);
        NODE_NCCL(ncclCommInitAll(comm.data(), parts, dev.data()));
        NODE_NCCL(ncclGroupStart());
        for (int r = 0; r < parts; ++r) {
            src[r] = d_stage + stage_off[r];
            if (!part_bytes[r]) continue;
            NODE_NCCL(ncclSend(d_strip[r], part_bytes[r], ncclUint8, 0, comm[r], stream[r]));
            NODE_NCCL(ncclRecv(d_stage + stage_off[r], part_bytes[r], ncclUint8, r, comm[0], stream[0]));
        }
        NODE_NCCL(ncclGroupEnd());
        for (int r = 0; r < parts; ++r) {
            NODE_HIP(hipSetDevice(dev[r]));
            NODE_HIP(hipStreamSynchronize(stream[r]));
        }
        for (auto c : comm) ncclCommDestroy(c);
        NODE_HIP(hipSetDevice(root));
    }
    // ... and band b of the image is band b / parts of part b % parts
    const int bands = (H + 7) / 8;
    for (int b = 0; b < bands; ++b) {
        const int rows = (H - b * 8 < 8) ? H - b * 8 : 8;
        NODE_HIP(hipMemcpyAsync(d_full + (size_t)b * 8 * W * 3, src[b % parts] + (size_t)(b / parts) * 8 * W * 3,
                                (size_t)rows * W * 3, hipMemcpyDeviceToDevice, stream[0]));
    }
    NODE_HIP(hipStreamSynchronize(stream[0]));
    NODE_HIP(hipSetDevice(root));
    NODE_HIP(hipMemcpy(out_u8_host, d_full, (size_t)W * H * 3, hipMemcpyDeviceToHost));
    (void)hipFree(d_full);
    if (d_stage) (void)hipFree(d_stage);
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
            // parts on different GPUs overlap in time: the frame's kernel time is the longest part;
            // virtual parts on one GPU run back to back
            total->kernel_ms = virtual_strips > 0 ? total->kernel_ms + stats[r].kernel_ms
                                                  : (stats[r].kernel_ms > total->kernel_ms ? stats[r].kernel_ms : total->kernel_ms);
        }
    }
    return RTM_OK;
}

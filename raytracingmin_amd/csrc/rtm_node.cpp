// rtm_node.cpp — one-process, N-GPU rendering for the host program: the image is dealt out in
// interleaved 8-row bands (band b goes to GPU b mod N: every GPU gets the same mix of cheap and costly
// rows), every GPU of the node renders its bands through the C ABI (rtm_render_scene with
// band_count/band_index, one host thread per device), and the band stacks are collected on device 0
// with ONE grouped RCCL exchange over xGMI (ncclSend from every device, matching ncclRecv on device 0),
// where the bands are put back in image order.  What travels is the float3 accumulation buffer
// (north_star) with the 8-bit view of the same rows behind it in the same message — the 8-bit image has
// to be quantised from the fp64 value (src/Renderer.cpp:253), which only the rendering GPU has.
// The reference has no multi-device code; this is the north_star's "image tiled across the 8 GPUs of
// one node with a single RCCL gather".  Linked into rtm_cli only — librtm_hip.so itself stays free of
// RCCL so that it can share a process with PyTorch's bundled copy.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
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

// Everything one part owns; released on every return path.
struct Part {
    int dev = 0;
    rtm_options opt{};
    size_t rows = 0;
    hipStream_t stream = nullptr;
    unsigned char* strip = nullptr;  // [rows*W*3 floats][rows*W*3 bytes]
    rtm_scene* scene = nullptr;
    rtm_stats stats{};
    int rc = RTM_OK;
    std::string detail;
    ~Part() {
        (void)hipSetDevice(dev);
        if (scene) (void)rtm_scene_destroy(scene);
        if (strip) (void)hipFree(strip);
        if (stream) (void)hipStreamDestroy(stream);
    }
};
struct RootBuffers {
    int dev = 0;
    unsigned char* stage = nullptr;
    float* full32 = nullptr;
    uint8_t* full8 = nullptr;
    ~RootBuffers() {
        (void)hipSetDevice(dev);
        if (stage) (void)hipFree(stage);
        if (full32) (void)hipFree(full32);
        if (full8) (void)hipFree(full8);
    }
};
struct Comms {
    std::vector<ncclComm_t> c;
    ~Comms() {
        for (auto v : c)
            if (v) (void)ncclCommDestroy(v);
    }
};
}  // namespace

int rtm_node_render(const rtm_settings* st, const rtm_object* objects, size_t n, const rtm_options* base,
                    int n_devices, int virtual_strips, int force_rccl, float* out_f32_host, uint8_t* out_u8_host,
                    rtm_stats* total, std::string& err) {
    if (!st || !base || (!out_u8_host && !out_f32_host) || n_devices < 1) return RTM_ERR_INVALID_ARGUMENT;
    const int W = st->width, H = st->height;
    const size_t px_bytes = 3 * sizeof(float) + 3;  // float3 + rgb8 per pixel
    // virtual_strips > 0: that many parts, all on base->device (exercises the tiling on one GPU)
    const int parts = virtual_strips > 0 ? virtual_strips : n_devices;
    std::vector<Part> part(parts);
    std::vector<size_t> stage_off(parts + 1, 0);
    for (int r = 0; r < parts; ++r) {
        Part& p = part[r];
        p.dev = virtual_strips > 0 ? base->device : r;
        p.opt = *base;
        p.opt.device = p.dev;
        p.opt.row_begin = 0;
        p.opt.row_end = H;
        p.opt.band_count = parts;
        p.opt.band_index = r;
        p.rows = (size_t)rtm_output_rows(&p.opt);
        stage_off[r + 1] = (stage_off[r] + p.rows * W * px_bytes + 15) & ~(size_t)15;
    }
    const int root = part[0].dev;
    for (int r = 0; r < parts; ++r) {
        Part& p = part[r];
        NODE_HIP(hipSetDevice(p.dev));
        NODE_HIP(hipStreamCreate(&p.stream));
        if (p.rows) NODE_HIP(hipMalloc((void**)&p.strip, p.rows * W * px_bytes));
        const int rc = rtm_scene_create_objects(objects, n, p.dev, &p.scene);
        if (rc != RTM_OK) {
            err = std::string("rtm_scene_create_objects: ") + rtm_last_error_detail();
            return rc;
        }
    }
    // one host thread per part: the renders of different GPUs run concurrently
    auto work = [&](int r) {
        Part& p = part[r];
        float* f32 = reinterpret_cast<float*>(p.strip);
        uint8_t* u8 = p.strip ? p.strip + p.rows * W * 3 * sizeof(float) : nullptr;
        p.rc = rtm_render_scene(st, p.scene, &p.opt, nullptr, f32, u8, p.stream, &p.stats);
        if (p.rc != RTM_OK) p.detail = rtm_last_error_detail();
    };
    if (virtual_strips > 0) {
        for (int r = 0; r < parts; ++r) work(r);
    } else {
        std::vector<std::thread> th;
        for (int r = 0; r < parts; ++r) th.emplace_back(work, r);
        for (auto& t : th) t.join();
    }
    for (int r = 0; r < parts; ++r)
        if (part[r].rc != RTM_OK) {
            err = "part " + std::to_string(r) + ": " + part[r].detail;
            return part[r].rc;
        }

    // gather on the root device: the band stacks land side by side in a staging buffer ...
    RootBuffers rb;
    rb.dev = root;
    NODE_HIP(hipSetDevice(root));
    NODE_HIP(hipMalloc((void**)&rb.full32, (size_t)W * H * 3 * sizeof(float)));
    NODE_HIP(hipMalloc((void**)&rb.full8, (size_t)W * H * 3));
    std::vector<const unsigned char*> src(parts);
    const bool use_rccl = virtual_strips <= 0 && (n_devices > 1 || force_rccl);
    if (!use_rccl) {
        for (int r = 0; r < parts; ++r) src[r] = part[r].strip;
    } else {
        NODE_HIP(hipMalloc((void**)&rb.stage, stage_off[parts] ? stage_off[parts] : 1));
        // one process, one node: RCCL's bootstrap needs no more than the loopback interface (a container's veth is
        // not always usable for it); an NCCL_SOCKET_IFNAME the user has set is respected
        (void)setenv("NCCL_SOCKET_IFNAME", "lo", 0);
        Comms comms;
        comms.c.assign(parts, nullptr);
        std::vector<int> devs(parts);
        for (int r = 0; r < parts; ++r) devs[r] = part[r].dev;
        NODE_NCCL(ncclCommInitAll(comms.c.data(), parts, devs.data()));
        NODE_NCCL(ncclGroupStart());
        for (int r = 0; r < parts; ++r) {
            src[r] = rb.stage + stage_off[r];
            const size_t bytes = part[r].rows * W * px_bytes;
            if (!bytes) continue;
            NODE_NCCL(ncclSend(part[r].strip, bytes, ncclUint8, 0, comms.c[r], part[r].stream));
            NODE_NCCL(ncclRecv(rb.stage + stage_off[r], bytes, ncclUint8, r, comms.c[0], part[0].stream));
        }
        NODE_NCCL(ncclGroupEnd());
        for (int r = 0; r < parts; ++r) {
            NODE_HIP(hipSetDevice(part[r].dev));
            NODE_HIP(hipStreamSynchronize(part[r].stream));
        }
        NODE_HIP(hipSetDevice(root));
    }
    // ... and band b of the image is band b / parts of part b % parts
    const int bands = (H + 7) / 8;
    for (int b = 0; b < bands; ++b) {
        const int rows = (H - b * 8 < 8) ? H - b * 8 : 8;
        const Part& p = part[b % parts];
        const unsigned char* s32 = src[b % parts] + (size_t)(b / parts) * 8 * W * 3 * sizeof(float);
        const unsigned char* s8 = src[b % parts] + p.rows * W * 3 * sizeof(float) + (size_t)(b / parts) * 8 * W * 3;
        NODE_HIP(hipMemcpyAsync(rb.full32 + (size_t)b * 8 * W * 3, s32, (size_t)rows * W * 3 * sizeof(float),
                                hipMemcpyDeviceToDevice, part[0].stream));
        NODE_HIP(hipMemcpyAsync(rb.full8 + (size_t)b * 8 * W * 3, s8, (size_t)rows * W * 3, hipMemcpyDeviceToDevice,
                                part[0].stream));
    }
    NODE_HIP(hipStreamSynchronize(part[0].stream));
    if (out_f32_host) NODE_HIP(hipMemcpy(out_f32_host, rb.full32, (size_t)W * H * 3 * sizeof(float), hipMemcpyDeviceToHost));
    if (out_u8_host) NODE_HIP(hipMemcpy(out_u8_host, rb.full8, (size_t)W * H * 3, hipMemcpyDeviceToHost));
    if (total) {
        std::memset(total, 0, sizeof *total);
        for (int r = 0; r < parts; ++r) {
            const rtm_stats& s = part[r].stats;
            total->samples += s.samples;
            total->casts += s.casts;
            total->bounces += s.bounces;
            total->draws += s.draws;
            // parts on different GPUs overlap in time: the frame's kernel time is the longest part;
            // virtual parts on one GPU run back to back
            total->kernel_ms = virtual_strips > 0 ? total->kernel_ms + s.kernel_ms
                                                  : (s.kernel_ms > total->kernel_ms ? s.kernel_ms : total->kernel_ms);
            total->variant = s.variant;
            total->split = s.split;
        }
    }
    return RTM_OK;
}

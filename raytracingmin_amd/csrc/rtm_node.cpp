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

#include <unistd.h>

#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
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

// Every step of the multi-GPU path announces itself on stderr (one write per line, never buffered), and a watchdog
// thread ends the process — a fresh non-zero exit, code 3, naming the stage — when a stage does not finish within its
// bound.  A run that stalls (a communicator that never comes up, a send/recv pair that never matches, a kernel that
// never drains) therefore says WHERE it stalled instead of dying silently at its caller's limit.
// Bounds: RTM_NODE_STAGE_TIMEOUT seconds for the set-up / RCCL / assembly stages (default 60), RTM_NODE_RENDER_TIMEOUT
// for the render stage (default 0 = unbounded: a frame may legitimately take minutes); RTM_NODE_QUIET=1 drops the lines
// (never the watchdog's).
class Stages {
  public:
    Stages() {
        stage_limit_ = env_seconds("RTM_NODE_STAGE_TIMEOUT", 60);
        render_limit_ = env_seconds("RTM_NODE_RENDER_TIMEOUT", 0);
        const char* q = std::getenv("RTM_NODE_QUIET");
        quiet_ = q && q[0] == '1';
        dog_ = std::thread([this] { watch(); });
    }
    ~Stages() {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
        }
        cv_.notify_all();
        dog_.join();
    }
    void enter(bool render_stage, const char* fmt, ...) __attribute__((format(printf, 3, 4))) {
        char buf[256];
        va_list ap;
        va_start(ap, fmt);
        std::vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        {
            std::lock_guard<std::mutex> g(m_);
            name_ = buf;
            since_ = std::chrono::steady_clock::now();
            limit_ = render_stage ? render_limit_ : stage_limit_;
        }
        say("stage: ", buf);
        // test hook: RTM_NODE_DEBUG_STALL=<text> parks the calling thread in the first stage whose name contains
        // <text>, which is how tests/test_cli_gpu.py shows that a stalled stage ends in exit code 3 with its name
        const char* stall = std::getenv("RTM_NODE_DEBUG_STALL");
        if (stall && *stall && std::strstr(buf, stall))
            for (;;) ::pause();
    }
    void note(const char* fmt, ...) __attribute__((format(printf, 2, 3))) {
        char buf[256];
        va_list ap;
        va_start(ap, fmt);
        std::vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        say("", buf);
    }

  private:
    static int env_seconds(const char* name, int dflt) {
        const char* v = std::getenv(name);
        return v && *v ? std::atoi(v) : dflt;
    }
    void say(const char* prefix, const char* text) const {
        if (quiet_) return;
        char line[320];
        int n = std::snprintf(line, sizeof line - 1, "rtm_node: %s%s", prefix, text);
        if (n > (int)sizeof line - 2) n = (int)sizeof line - 2;
        line[n++] = '\n';
        (void)!::write(2, line, (size_t)n);  // one write(2) per line: nothing to flush, nothing to interleave
    }
    void watch() {
        std::unique_lock<std::mutex> g(m_);
        while (!stop_) {
            cv_.wait_for(g, std::chrono::milliseconds(500));
            if (stop_ || limit_ <= 0 || name_.empty()) continue;
            const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - since_).count();
            if (waited > limit_) {
                char line[400];
                const int n = std::snprintf(line, sizeof line,
                                            "rtm_node: WATCHDOG: stage '%s' has not finished after %d s; exiting with code 3\n",
                                            name_.c_str(), limit_);
                (void)!::write(2, line, (size_t)n);
                ::_exit(3);
            }
        }
    }
    std::mutex m_;
    std::condition_variable cv_;
    std::string name_;
    std::chrono::steady_clock::time_point since_{};
    int limit_ = 0, stage_limit_ = 60, render_limit_ = 0;
    bool stop_ = false, quiet_ = false;
    std::thread dog_;
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
    Stages stages;  // first, so that it outlives (and watches) the release of everything below
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
        stages.enter(false, "set-up of part %d of %d on device %d (stream, band stack, scene upload)", r, parts, p.dev);
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
        stages.note("part %d rendered on device %d: rc %d, %.3f ms of kernels", r, p.dev, p.rc, p.stats.kernel_ms);
    };
    stages.enter(true, "render of %d part(s): %dx%d, %zu row(s) in the largest", parts, W, H, part[0].rows);
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
    stages.enter(false, "frame buffers on the root device %d", root);
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
        // not always usable for it); rtm_cli's main() sets NCCL_SOCKET_IFNAME=lo before anything else runs unless the
        // user has set one — here, with HIP threads alive, the environment is only read
        const char* ifname = std::getenv("NCCL_SOCKET_IFNAME");
        Comms comms;
        comms.c.assign(parts, nullptr);
        std::vector<int> devs(parts);
        for (int r = 0; r < parts; ++r) devs[r] = part[r].dev;
        stages.enter(false, "ncclCommInitAll over %d device(s), bootstrap interface %s", parts, ifname ? ifname : "(RCCL's choice)");
        NODE_NCCL(ncclCommInitAll(comms.c.data(), parts, devs.data()));
        stages.enter(false, "grouped ncclSend/ncclRecv of %zu bytes into device %d", stage_off[parts], root);
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
            stages.enter(false, "stream synchronise after the exchange, part %d on device %d", r, part[r].dev);
            NODE_HIP(hipSetDevice(part[r].dev));
            NODE_HIP(hipStreamSynchronize(part[r].stream));
        }
        NODE_HIP(hipSetDevice(root));
        stages.enter(false, "ncclCommDestroy of %d communicator(s)", parts);
    }
    // ... and band b of the image is band b / parts of part b % parts
    const int bands = (H + 7) / 8;
    stages.enter(false, "de-interleave of %d band(s) on the root device + copy to the host", bands);
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
    stages.enter(false, "release of the parts' scenes, buffers and streams");
    return RTM_OK;
}

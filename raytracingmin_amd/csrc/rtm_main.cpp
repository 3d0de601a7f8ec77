// rtm_main.cpp — host program: the reference's main() (src/main.cpp:8-46) on top of the C ABI.
//
//   rtm_cli [-?] [-json <file>] [-sampleJson]            (the reference's flags, same defaults)
//           [--width N] [--height N] [--samples N] [--superSamples N] [--spp N]
//           [--mode literal|repaired] [--max-bounces N] [--seed N] [--device N] [--out STEM] [--device-trig]
//           [--gpus N] [--virtual-strips N] [--force-rccl]   (interleaved 8-row bands over N GPUs + one RCCL gather)
//           [--dump-f32 FILE]                    (the gathered float3 buffer, raw little-endian floats)
//
// Flow of the reference: pick the JSON (default settingData.json), create the sample JSON when it
// does not exist, load, render, write <stem>.jpg (quality 60) and <stem>.bmp with stem "result".
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rtm.h"
#include "rtm_node.h"

static bool file_exists(const std::string& p) {
    FILE* f = std::fopen(p.c_str(), "rb");
    if (f) std::fclose(f);
    return f != nullptr;
}

static void usage() {
    std::printf(
        "Usage: rtm_cli [OPTION]...\n\n"
        "-sampleJson : save the sample scene json file as settingData.json\n"
        "-json <file> : scene json file; settingData.json when not given\n"
        "--width/--height/--samples/--superSamples N : override the file's values\n"
        "--spp N : samples = N / superSamples^2\n"
        "--mode literal|repaired (default repaired), --max-bounces N (default -1 = unlimited)\n"
        "--seed N, --device N, --out STEM (default result)\n"
        "--device-trig : the device's own sin/cos instead of this host's libm values (default: host values, bit-identical\n"
        "              to a CPU run of the reference even where deep paths amplify one-ulp differences; device-trig is ~2 %% faster)\n"
        "--gpus N : interleaved 8-row bands over N GPUs of this node, one RCCL gather of the float3 buffer;\n"
        "--force-rccl : take the RCCL exchange with --gpus 1 too; --virtual-strips N : N parts on one GPU, no RCCL\n"
        "--dump-f32 FILE : write the float3 accumulation buffer (raw floats, row-major RGB)\n");
}

int main(int argc, char* argv[]) {
    // progress lines reach a pipe as they are printed (a caller that has to kill a stalled run still sees how far it got)
    std::setvbuf(stdout, nullptr, _IOLBF, 0);
    // N GPUs of ONE node in ONE process: RCCL's bootstrap needs no more than the loopback interface, and a
    // container's veth is not always connectable to itself.  Set before any thread exists; a user's value is kept.
    (void)setenv("NCCL_SOCKET_IFNAME", "lo", 0);
    std::string json_file = "settingData.json", stem = "result";
    int width = 0, height = 0, samples = 0, super_samples = 0, spp = 0;
    int mode = RTM_MODE_REPAIRED, max_bounces = -1, device = 0, gpus = 1, virtual_strips = 0, host_trig = 1, force_rccl = 0;
    std::string dump_f32;
    unsigned long long seed = 0x5EED;
    for (int i = 1; i < argc; ++i) {
        const std::string c = argv[i];
        auto next_int = [&](int& dst) {
            if (i + 1 < argc) dst = std::atoi(argv[++i]);
        };
        if (c == "-?") {
            usage();
            return 0;
        } else if (c == "-json") {  // src/main.cpp:18-28: falls back when the value is missing or a flag
            if (argc <= i + 1 || argv[i + 1][0] == '-')
                json_file = "settingData.json";
            else
                json_file = argv[++i];
        } else if (c == "-sampleJson") {  // src/main.cpp:29-33
            std::printf("saving the sample scene json file: settingData.json\n");
            return rtm_scene_save_sample_json("settingData.json") == RTM_OK ? 0 : 1;
        } else if (c == "--width") next_int(width);
        else if (c == "--height") next_int(height);
        else if (c == "--samples") next_int(samples);
        else if (c == "--superSamples") next_int(super_samples);
        else if (c == "--spp") next_int(spp);
        else if (c == "--max-bounces") next_int(max_bounces);
        else if (c == "--device") next_int(device);
        else if (c == "--gpus") next_int(gpus);
        else if (c == "--virtual-strips") next_int(virtual_strips);
        else if (c == "--host-trig") host_trig = 1;
        else if (c == "--device-trig") host_trig = 0;
        else if (c == "--force-rccl") force_rccl = 1;
        else if (c == "--dump-f32" && i + 1 < argc) dump_f32 = argv[++i];
        else if (c == "--seed" && i + 1 < argc) seed = std::strtoull(argv[++i], nullptr, 0);
        else if (c == "--out" && i + 1 < argc) stem = argv[++i];
        else if (c == "--mode" && i + 1 < argc) {
            const std::string m = argv[++i];
            if (m == "literal") mode = RTM_MODE_LITERAL;
            else if (m == "repaired") mode = RTM_MODE_REPAIRED;
            else {
                std::fprintf(stderr, "unknown mode %s\n", m.c_str());
                return 2;
            }
        }
    }
    if (!file_exists(json_file)) {  // src/main.cpp:36-39
        std::printf("saving the sample scene json file: %s\n", json_file.c_str());
        if (rtm_scene_save_sample_json(json_file.c_str()) != RTM_OK) return 1;
    }
    std::printf("loading %s and starting the render\n", json_file.c_str());  // src/main.cpp:40

    rtm_settings st;
    size_t n = 0;
    const int literal = (mode == RTM_MODE_LITERAL);
    // the any-type object list: the shipped files hold spheres only; objectType 2 (png::PlaneObject) loads too
    int rc = rtm_scene_load_json_objects(json_file.c_str(), literal, &st, nullptr, 0, &n);
    if (rc != RTM_OK) {
        std::fprintf(stderr, "%s: %s (%s)\n", json_file.c_str(), rtm_strerror(rc), rtm_last_error_detail());
        return 1;
    }
    std::vector<rtm_object> spheres(n ? n : 1);
    rc = rtm_scene_load_json_objects(json_file.c_str(), literal, &st, spheres.data(), n, &n);
    if (rc != RTM_OK) {
        std::fprintf(stderr, "%s: %s (%s)\n", json_file.c_str(), rtm_strerror(rc), rtm_last_error_detail());
        return 1;
    }
    if (width > 0) st.width = width;
    if (height > 0) st.height = height;
    if (super_samples > 0) st.super_samples = super_samples;
    if (samples > 0) st.samples = samples;
    if (spp > 0) st.samples = spp / (st.super_samples * st.super_samples) > 0 ? spp / (st.super_samples * st.super_samples) : 1;

    rtm_options opt;
    std::memset(&opt, 0, sizeof opt);
    opt.mode = mode | (host_trig ? RTM_MODE_HOST_TRIG : 0);
    opt.max_bounces = max_bounces;
    opt.seed = seed;
    opt.row_begin = 0;
    opt.row_end = st.height;
    opt.device = device;

    const size_t vals = (size_t)st.width * st.height * 3;
    std::vector<uint8_t> rgb8(vals);
    std::vector<float> rgb32(dump_f32.empty() ? 0 : vals);
    rtm_stats stats;
    if (gpus > 1 || virtual_strips > 0 || force_rccl) {
        int have = 0;
        if (rtm_device_count(&have) != RTM_OK || have < gpus) {
            std::fprintf(stderr, "--gpus %d requested, %d HIP device(s) present\n", gpus, have);
            return 1;
        }
        std::string err;
        rc = rtm_node_render(&st, spheres.data(), n, &opt, gpus, virtual_strips, force_rccl,
                             rgb32.empty() ? nullptr : rgb32.data(), rgb8.data(), &stats, err);
        if (rc != RTM_OK) {
            std::fprintf(stderr, "render failed: %s (%s)\n", rtm_strerror(rc), err.c_str());
            return 1;
        }
    } else {
        rc = rtm_render_objects(&st, spheres.data(), n, &opt, nullptr, rgb32.empty() ? nullptr : rgb32.data(), rgb8.data(),
                                &stats);
        if (rc != RTM_OK) {
            std::fprintf(stderr, "render failed: %s (%s)\n", rtm_strerror(rc), rtm_last_error_detail());
            return 1;
        }
    }
    if (!dump_f32.empty()) {
        FILE* f = std::fopen(dump_f32.c_str(), "wb");
        if (!f || std::fwrite(rgb32.data(), sizeof(float), vals, f) != vals) {
            std::fprintf(stderr, "cannot write %s\n", dump_f32.c_str());
            return 1;
        }
        std::fclose(f);
    }
    std::printf("%d x %d, %llu samples, %.3f casts/sample, kernel %.3f ms, %.1f Msamples/s\n", st.width,
                st.height, (unsigned long long)stats.samples,
                stats.samples ? (double)stats.casts / (double)stats.samples : 0.0, stats.kernel_ms,
                stats.kernel_ms > 0 ? (double)stats.samples / stats.kernel_ms * 1e-3 : 0.0);
    // src/Renderer.cpp:256-257
    const int ok_jpg = rtm_write_jpg((stem + ".jpg").c_str(), st.width, st.height, 3, rgb8.data(), 60);
    const int ok_bmp = rtm_write_bmp((stem + ".bmp").c_str(), st.width, st.height, 3, rgb8.data());
    return (ok_jpg && ok_bmp) ? 0 : 1;
}

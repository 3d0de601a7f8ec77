// rtm_image.cpp — host-side image output: the reference's 8-bit quantisation and the two files it
// writes through stb_image_write (reference src/Renderer.cpp:251-257).  stb is an empty submodule
// in the checkout and absent from the image, so these are own writers behind stb-compatible
// signatures (int return, 1 = success): a 24-bit bottom-up BGR BMP with stb's 54-byte header, and a
// baseline JFIF JPEG (standard Annex-K tables scaled by quality, 4:2:0 at quality <= 90 like
// stb's encoder).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "rtm_internal.h"

namespace rtm {

// src/Renderer.cpp:253: resultImage[i] = (unsigned char)255 * std::min(image[i], 1.0);
// i.e. int 255 times a double, truncated on the store to unsigned char.  No gamma.
int quantise(const double* image, size_t n_values, uint8_t* out) {
    if ((!image || !out) && n_values) {
        set_last_error("null argument");
        return RTM_ERR_INVALID_ARGUMENT;
    }
    for (size_t i = 0; i < n_values; ++i) {
        const double m = (1.0 < image[i]) ? 1.0 : image[i];  // std::min(a, b) = (b < a) ? b : a
        const double v = 255 * m;
        out[i] = (v >= 0.0 && v < 256.0) ? (uint8_t)v : (uint8_t)0;  // out of range is UB upstream
    }
    return RTM_OK;
}

namespace {
struct File {
    FILE* f;
    explicit File(const char* name) : f(name ? std::fopen(name, "wb") : nullptr) {}
    ~File() {
        if (f) std::fclose(f);
    }
    void u8(unsigned v) { std::fputc((int)(v & 0xFF), f); }
    void le16(unsigned v) {
        u8(v);
        u8(v >> 8);
    }
    void le32(unsigned v) {
        le16(v);
        le16(v >> 16);
    }
    void be16(unsigned v) {
        u8(v >> 8);
        u8(v);
    }
    bool ok() const { return f && !std::ferror(f); }
};
}  // namespace

// stbi_write_bmp layout for comp = 3: "BM", file size, 0, 0, data offset 54; BITMAPINFOHEADER
// (40, w, h, 1 plane, 24 bpp, no compression, zeros); rows bottom-up, BGR, padded to 4 bytes.
int write_bmp(const char* filename, int w, int h, int comp, const void* data) {
    if (!filename || !data || w <= 0 || h <= 0 || (comp != 3 && comp != 1 && comp != 4)) return 0;
    File out(filename);
    if (!out.f) return 0;
    const unsigned pad = (unsigned)(-w * 3) & 3u;
    out.u8('B');
    out.u8('M');
    out.le32(14u + 40u + ((unsigned)w * 3u + pad) * (unsigned)h);
    out.le16(0);
    out.le16(0);
    out.le32(14 + 40);
    out.le32(40);
    out.le32((unsigned)w);
    out.le32((unsigned)h);
    out.le16(1);
    out.le16(24);
    for (int k = 0; k < 6; ++k) out.le32(0);
    const uint8_t* px = (const uint8_t*)data;
    for (int y = h - 1; y >= 0; --y) {
        const uint8_t* row = px + (size_t)y * w * comp;
        for (int x = 0; x < w; ++x) {
            const uint8_t* p = row + (size_t)x * comp;
            if (comp == 1) {
                out.u8(p[0]);
                out.u8(p[0]);
                out.u8(p[0]);
            } else {
                out.u8(p[2]);
                out.u8(p[1]);
                out.u8(p[0]);
            }
        }
        for (unsigned k = 0; k < pad; ++k) out.u8(0);
    }
    return out.ok() ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
// Baseline JPEG
namespace {
const uint8_t kZigZag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
                             12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                             35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
                             58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
// ITU-T T.81 Annex K.1, natural order
const uint8_t kLumaQ[64] = {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,
                            14, 13, 16, 24, 40,  57,  69,  56,  14, 17, 22, 29, 51,  87,  80,  62,
                            18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
                            49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
const uint8_t kChromaQ[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99,
                              24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
                              99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                              99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
// ITU-T T.81 Annex K.3 Huffman specifications: 16 code-length counts, then the symbols
const uint8_t kDcLumaBits[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
const uint8_t kDcChromaBits[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
const uint8_t kDcVals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
const uint8_t kAcLumaBits[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
const uint8_t kAcLumaVals[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07,
    0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0,
    0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28,
    0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49,
    0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69,
    0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89,
    0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7,
    0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5,
    0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
    0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8,
    0xf9, 0xfa};
const uint8_t kAcChromaBits[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
const uint8_t kAcChromaVals[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71,
    0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0,
    0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26,
    0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48,
    0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68,
    0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87,
    0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5,
    0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
    0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8,
    0xf9, 0xfa};

struct HuffTable {
    uint16_t code[256];
    uint8_t len[256];
    void build(const uint8_t bits[16], const uint8_t* vals) {
        std::memset(len, 0, sizeof len);
        unsigned c = 0, k = 0;
        for (int l = 1; l <= 16; ++l) {
            for (int i = 0; i < bits[l - 1]; ++i, ++k) {
                code[vals[k]] = (uint16_t)c++;
                len[vals[k]] = (uint8_t)l;
            }
            c <<= 1;
        }
    }
};

struct BitWriter {
    File& out;
    unsigned acc = 0;
    int n = 0;
    explicit BitWriter(File& f) : out(f) {}
    void put(unsigned code, int len) {
        acc = (acc << len) | (code & ((1u << len) - 1u));
        n += len;
        while (n >= 8) {
            const unsigned b = (acc >> (n - 8)) & 0xFF;
            out.u8(b);
            if (b == 0xFF) out.u8(0);  // byte stuffing
            n -= 8;
        }
    }
    void flush() {
        if (n > 0) put(0x7F, 8 - n);  // pad with ones
    }
};

// 8x8 forward DCT-II (separable, orthonormal scaling of T.81 A.3.3)
void fdct8x8(const float in[64], float out[64]) {
    static float ct[8][8];
    static bool init = false;
    if (!init) {
        for (int u = 0; u < 8; ++u)
            for (int x = 0; x < 8; ++x)
                ct[u][x] = (float)((u == 0 ? std::sqrt(0.125) : 0.5) * std::cos((2 * x + 1) * u * M_PI / 16.0));
        init = true;
    }
    float tmp[64];
    for (int y = 0; y < 8; ++y)
        for (int u = 0; u < 8; ++u) {
            float s = 0.f;
            for (int x = 0; x < 8; ++x) s += in[y * 8 + x] * ct[u][x];
            tmp[y * 8 + u] = s;
        }
    for (int v = 0; v < 8; ++v)
        for (int u = 0; u < 8; ++u) {
            float s = 0.f;
            for (int y = 0; y < 8; ++y) s += tmp[y * 8 + u] * ct[v][y];
            out[v * 8 + u] = s;
        }
}

void size_and_bits(int v, int& nbits, unsigned& bits) {
    int a = v < 0 ? -v : v;
    nbits = 0;
    while (a) {
        ++nbits;
        a >>= 1;
    }
    bits = (unsigned)(v < 0 ? v - 1 : v) & ((1u << nbits) - 1u);
}

int encode_block(BitWriter& bw, const float px[64], const uint8_t q[64], int prev_dc,
                 const HuffTable& dc, const HuffTable& ac) {
    float f[64];
    fdct8x8(px, f);
    int zz[64];
    for (int i = 0; i < 64; ++i) {
        const float v = f[kZigZag[i]] / (float)q[kZigZag[i]];
        zz[i] = (int)(v < 0 ? v - 0.5f : v + 0.5f);
    }
    int nbits;
    unsigned bits;
    size_and_bits(zz[0] - prev_dc, nbits, bits);
    bw.put(dc.code[nbits], dc.len[nbits]);
    if (nbits) bw.put(bits, nbits);
    int last = 63;
    while (last > 0 && zz[last] == 0) --last;
    int run = 0;
    for (int i = 1; i <= last; ++i) {
        if (zz[i] == 0) {
            ++run;
            continue;
        }
        while (run >= 16) {
            bw.put(ac.code[0xF0], ac.len[0xF0]);  // ZRL
            run -= 16;
        }
        size_and_bits(zz[i], nbits, bits);
        const int sym = (run << 4) | nbits;
        bw.put(ac.code[sym], ac.len[sym]);
        bw.put(bits, nbits);
        run = 0;
    }
    if (last != 63) bw.put(ac.code[0x00], ac.len[0x00]);  // EOB
    return zz[0];
}
}  // namespace

int write_jpg(const char* filename, int w, int h, int comp, const void* data, int quality) {
    if (!filename || !data || w <= 0 || h <= 0 || w > 65535 || h > 65535 || (comp != 3 && comp != 1 && comp != 4))
        return 0;
    File out(filename);
    if (!out.f) return 0;
    quality = quality ? quality : 90;
    quality = quality < 1 ? 1 : (quality > 100 ? 100 : quality);
    const bool subsample = quality <= 90;
    const int scale = quality < 50 ? 5000 / quality : 200 - quality * 2;
    uint8_t qy[64], qc[64];
    for (int i = 0; i < 64; ++i) {
        int a = (kLumaQ[i] * scale + 50) / 100, b = (kChromaQ[i] * scale + 50) / 100;
        qy[i] = (uint8_t)(a < 1 ? 1 : (a > 255 ? 255 : a));
        qc[i] = (uint8_t)(b < 1 ? 1 : (b > 255 ? 255 : b));
    }
    HuffTable dcY, dcC, acY, acC;
    dcY.build(kDcLumaBits, kDcVals);
    dcC.build(kDcChromaBits, kDcVals);
    acY.build(kAcLumaBits, kAcLumaVals);
    acC.build(kAcChromaBits, kAcChromaVals);

    // SOI, APP0 (JFIF 1.1, no density), DQT x2
    out.be16(0xFFD8);
    out.be16(0xFFE0);
    out.be16(16);
    for (const char* c = "JFIF"; *c; ++c) out.u8((unsigned)*c);
    out.u8(0);
    out.be16(0x0101);
    out.u8(0);
    out.be16(1);
    out.be16(1);
    out.u8(0);
    out.u8(0);
    out.be16(0xFFDB);
    out.be16(2 + 65 * 2);
    out.u8(0);
    for (int i = 0; i < 64; ++i) out.u8(qy[kZigZag[i]]);
    out.u8(1);
    for (int i = 0; i < 64; ++i) out.u8(qc[kZigZag[i]]);
    // SOF0
    out.be16(0xFFC0);
    out.be16(17);
    out.u8(8);
    out.be16((unsigned)h);
    out.be16((unsigned)w);
    out.u8(3);
    out.u8(1);
    out.u8(subsample ? 0x22 : 0x11);
    out.u8(0);
    out.u8(2);
    out.u8(0x11);
    out.u8(1);
    out.u8(3);
    out.u8(0x11);
    out.u8(1);
    // DHT x4
    auto dht = [&](unsigned id, const uint8_t bits[16], const uint8_t* vals, int nvals) {
        out.be16(0xFFC4);
        out.be16((unsigned)(2 + 1 + 16 + nvals));
        out.u8(id);
        for (int i = 0; i < 16; ++i) out.u8(bits[i]);
        for (int i = 0; i < nvals; ++i) out.u8(vals[i]);
    };
    dht(0x00, kDcLumaBits, kDcVals, 12);
    dht(0x10, kAcLumaBits, kAcLumaVals, 162);
    dht(0x01, kDcChromaBits, kDcVals, 12);
    dht(0x11, kAcChromaBits, kAcChromaVals, 162);
    // SOS
    out.be16(0xFFDA);
    out.be16(12);
    out.u8(3);
    out.u8(1);
    out.u8(0x00);
    out.u8(2);
    out.u8(0x11);
    out.u8(3);
    out.u8(0x11);
    out.u8(0);
    out.u8(63);
    out.u8(0);

    const uint8_t* px = (const uint8_t*)data;
    auto ycc = [&](int x, int y, float& Y, float& U, float& V) {
        x = x < w ? x : w - 1;  // replicate the edge
        y = y < h ? y : h - 1;
        const uint8_t* p = px + ((size_t)y * w + x) * comp;
        const float r = p[0], g = comp == 1 ? p[0] : p[1], b = comp == 1 ? p[0] : p[2];
        Y = 0.299f * r + 0.587f * g + 0.114f * b - 128.f;
        U = -0.168736f * r - 0.331264f * g + 0.5f * b;
        V = 0.5f * r - 0.418688f * g - 0.081312f * b;
    };
    BitWriter bw(out);
    int dcy = 0, dcu = 0, dcv = 0;
    const int mcu = subsample ? 16 : 8;
    std::vector<float> Yb((size_t)mcu * mcu), Ub((size_t)mcu * mcu), Vb((size_t)mcu * mcu);
    for (int my = 0; my < h; my += mcu)
        for (int mx = 0; mx < w; mx += mcu) {
            for (int yy = 0; yy < mcu; ++yy)
                for (int xx = 0; xx < mcu; ++xx)
                    ycc(mx + xx, my + yy, Yb[(size_t)yy * mcu + xx], Ub[(size_t)yy * mcu + xx], Vb[(size_t)yy * mcu + xx]);
            float blk[64], ub[64], vb[64];
            if (subsample) {
                for (int by = 0; by < 2; ++by)
                    for (int bx = 0; bx < 2; ++bx) {
                        for (int yy = 0; yy < 8; ++yy)
                            for (int xx = 0; xx < 8; ++xx)
                                blk[yy * 8 + xx] = Yb[(size_t)(by * 8 + yy) * 16 + bx * 8 + xx];
                        dcy = encode_block(bw, blk, qy, dcy, dcY, acY);
                    }
                for (int yy = 0; yy < 8; ++yy)
                    for (int xx = 0; xx < 8; ++xx) {
                        const size_t i = (size_t)(yy * 2) * 16 + xx * 2;
                        ub[yy * 8 + xx] = (Ub[i] + Ub[i + 1] + Ub[i + 16] + Ub[i + 17]) * 0.25f;
                        vb[yy * 8 + xx] = (Vb[i] + Vb[i + 1] + Vb[i + 16] + Vb[i + 17]) * 0.25f;
                    }
                dcu = encode_block(bw, ub, qc, dcu, dcC, acC);
                dcv = encode_block(bw, vb, qc, dcv, dcC, acC);
            } else {
                dcy = encode_block(bw, Yb.data(), qy, dcy, dcY, acY);
                dcu = encode_block(bw, Ub.data(), qc, dcu, dcC, acC);
                dcv = encode_block(bw, Vb.data(), qc, dcv, dcC, acC);
            }
        }
    bw.flush();
    out.be16(0xFFD9);
    return out.ok() ? 1 : 0;
}

}  // namespace rtm

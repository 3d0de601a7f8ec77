// rtm_device.h — device-side math of the path-tracing hot path, fp64 with the reference's float
// islands.  Compiled with -ffp-contract=off: every a*b+c below is a separate multiply and add, in
// the operand order of the reference (citations: file:line in the reference checkout).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rtm {

struct D3 {
    double x, y, z;
};

__device__ __forceinline__ D3 d3(double x, double y, double z) { return D3{x, y, z}; }
// src/Ray.h:15-35
__device__ __forceinline__ D3 operator+(D3 a, D3 b) { return D3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ D3 operator-(D3 a, D3 b) { return D3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ D3 operator*(D3 a, D3 b) { return D3{a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ D3 operator*(D3 a, double s) { return D3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ D3 operator/(D3 a, double s) { return D3{a.x / s, a.y / s, a.z / s}; }

// src/Ray.h:61-63
__device__ __forceinline__ double dot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// src/Ray.h:64-66 (middle component is (-a.x)*b.z + a.z*b.x)
__device__ __forceinline__ D3 cross(D3 a, D3 b) {
    return D3{a.y * b.z - a.z * b.y, -a.x * b.z + a.z * b.x, a.x * b.y - a.y * b.x};
}
// src/Ray.h:67-69: std::sqrtf of a double => round to float, correctly rounded float sqrt, widen.
// hipcc's sqrtf is the correctly rounded expansion by default
// (-fhip-fp32-correctly-rounded-divide-sqrt); checked exhaustively by tests/test_device_math.py.
__device__ __forceinline__ double magnitude(D3 a) {
    const float len2 = (float)(a.x * a.x + a.y * a.y + a.z * a.z);
    return (double)__builtin_sqrtf(len2);
}
// src/Ray.h:70-72: three true divisions
__device__ __forceinline__ D3 normalize(D3 a) { return a / magnitude(a); }

// ---- build-defined counter RNG; must agree with rtm_rng_u01 on the host ----
__host__ __device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x21f0aaadu;
    x ^= x >> 15;
    x *= 0x735a2d97u;
    x ^= x >> 15;
    return x;
}
__host__ __device__ __forceinline__ uint64_t smfin64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ uint64_t seed_multiplier(uint64_t seed) {
    return smfin64(seed + 0x9E3779B97F4A7C15ull) | 1ull;
}
// Keying: one 64-bit mix per PIXEL, one 32-bit mix per SAMPLE, two 32-bit mixes per DRAW.
//   pixel key   (p0, p1) = halves of smfin64((pixel + 1) * seed_mult)
//   sample key  k0 = mix32(p0 + sample * 0x9E3779B9), k1 = p1 ^ (sample * 0x85EBCA6B)
//   draw j      bits = mix32(mix32(k0 + j * 0x9E3779B9) ^ k1);  u = (2 * (bits >> 9) + 1) * 2^-24
struct RngPixelKey {
    uint32_t p0, p1;
};
struct RngStream {
    uint32_t ctr, k1;  // ctr = k0 + index * 0x9E3779B9
};
__host__ __device__ __forceinline__ RngPixelKey rng_pixel_key(uint64_t seed_mult, uint32_t pixel) {
    const uint64_t z = smfin64(((uint64_t)pixel + 1ull) * seed_mult);
    return RngPixelKey{(uint32_t)z, (uint32_t)(z >> 32)};
}
__host__ __device__ __forceinline__ RngStream rng_open(RngPixelKey pk, uint32_t sample) {
    return RngStream{mix32(pk.p0 + sample * 0x9E3779B9u), pk.p1 ^ (sample * 0x85EBCA6Bu)};
}
__host__ __device__ __forceinline__ double rng_bits_to_u01(uint32_t x) {
    // 23 random bits -> odd multiple of 2^-24: never 0 or 1, exact in fp32 and fp64
    return (double)(2u * (x >> 9) + 1u) * (1.0 / 16777216.0);
}
__host__ __device__ __forceinline__ double rng_next(RngStream& s) {
    const uint32_t x = mix32(mix32(s.ctr) ^ s.k1);
    s.ctr += 0x9E3779B9u;
    return rng_bits_to_u01(x);
}
__host__ __device__ __forceinline__ double rng_u01_at(uint64_t seed_mult, uint32_t pixel, uint32_t sample,
                                                      uint32_t index) {
    RngStream s = rng_open(rng_pixel_key(seed_mult, pixel), sample);
    s.ctr += index * 0x9E3779B9u;
    return rng_next(s);
}

}  // namespace rtm

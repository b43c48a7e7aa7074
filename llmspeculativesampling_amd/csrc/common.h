// Shared helpers for libspecdec (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/specdec.h"

typedef __bf16 bf16_t;
typedef _Float16 f16_t;

void sd_set_error(const char *fmt, ...);

#define SD_HIP_CHECK(expr)                                                             \
    do {                                                                               \
        hipError_t _e = (expr);                                                        \
        if (_e != hipSuccess) {                                                        \
            sd_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return SD_ERR_HIP;                                                         \
        }                                                                              \
    } while (0)

#define SD_REQUIRE(cond, ...)                                                          \
    do {                                                                               \
        if (!(cond)) {                                                                 \
            sd_set_error(__VA_ARGS__);                                                 \
            return SD_ERR_INVALID;                                                     \
        }                                                                              \
    } while (0)

#define SD_LAUNCH_CHECK() SD_HIP_CHECK(hipGetLastError())

// ---- activation dtype helpers: T is float, bf16_t or f16_t.  rnd<T>(x) rounds an fp32 value to the
// storage type and back, which is how every per-op result of a bf16 reference model is rounded.
__device__ __forceinline__ float to_f(float x) { return x; }
__device__ __forceinline__ float to_f(bf16_t x) { return (float)x; }
__device__ __forceinline__ float to_f(f16_t x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f(float x);
template <> __device__ __forceinline__ float from_f<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_f<bf16_t>(float x) { return (bf16_t)x; }
template <> __device__ __forceinline__ f16_t from_f<f16_t>(float x) { return (f16_t)x; }
template <typename T> __device__ __forceinline__ float rnd(float x) { return to_f(from_f<T>(x)); }

// ---- wave / block reductions (wave = 64 lanes)
// Data-parallel-primitive moves instead of __shfl_xor (which hipcc lowers to six dependent ds_bpermute_b32 through the
// LDS crossbar, ~0.3 us per reduction): xor 1, xor 2, half-row mirror, row mirror leave every lane with its row's
// (16 lanes) result; row_bcast15 / row_bcast31 carry it across the four rows into lanes 48..63; lane 63 is read back.
// The order of the additions is fixed, so results stay bit-reproducible.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov(float v, float fill) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill), __builtin_bit_cast(int, v),
                                                                 CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_mov<0xB1, 0xF>(v, 0.f);        // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E, 0xF>(v, 0.f);        // quad_perm [2,3,0,1]
    v += dpp_mov<0x141, 0xF>(v, 0.f);       // row_half_mirror
    v += dpp_mov<0x140, 0xF>(v, 0.f);       // row_mirror
    v += dpp_mov<0x142, 0xA>(v, 0.f);       // row_bcast15 into rows 1 and 3
    v += dpp_mov<0x143, 0xC>(v, 0.f);       // row_bcast31 into rows 2 and 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1, 0xF>(v, v));
    v = fmaxf(v, dpp_mov<0x4E, 0xF>(v, v));
    v = fmaxf(v, dpp_mov<0x141, 0xF>(v, v));
    v = fmaxf(v, dpp_mov<0x140, 0xF>(v, v));
    v = fmaxf(v, dpp_mov<0x142, 0xA>(v, v));
    v = fmaxf(v, dpp_mov<0x143, 0xC>(v, v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// Sums / maxima over the two 32-lane halves of a wave (lanes 0-31 -> lo, 32-63 -> hi), every lane gets both.
// lo and hi are each "second row + first row" of their half; an inactive half yields an unspecified value.
__device__ __forceinline__ void half_sums(float v, float &lo, float &hi) {
    v += dpp_mov<0xB1, 0xF>(v, 0.f);
    v += dpp_mov<0x4E, 0xF>(v, 0.f);
    v += dpp_mov<0x141, 0xF>(v, 0.f);
    v += dpp_mov<0x140, 0xF>(v, 0.f);
    v += dpp_mov<0x142, 0xA>(v, 0.f);
    lo = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 31));
    hi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ void half_maxes(float v, float &lo, float &hi) {
    v = fmaxf(v, dpp_mov<0xB1, 0xF>(v, v));
    v = fmaxf(v, dpp_mov<0x4E, 0xF>(v, v));
    v = fmaxf(v, dpp_mov<0x141, 0xF>(v, v));
    v = fmaxf(v, dpp_mov<0x140, 0xF>(v, v));
    v = fmaxf(v, dpp_mov<0x142, 0xA>(v, v));
    lo = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 31));
    hi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Block-wide sum / max through a small LDS array (>= 16 floats); all threads get the result.
__device__ __forceinline__ float block_sum(float v, float *sh) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += sh[i];
    return r;
}
__device__ __forceinline__ float block_max(float v, float *sh) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_max(v);
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    float r = sh[0];
    for (int i = 1; i < nw; ++i) r = fmaxf(r, sh[i]);
    return r;
}

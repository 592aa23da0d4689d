// Memory-system micro-benchmarks: the denominators bench.py prices the step's kernels against
// ("measure achievable with a device memcpy/stream kernel and report both").
//   stream copy   -- float4 copy, one contiguous slice per workgroup (the optimiser kernels' walk)
//   row gather    -- vs_gather_mean itself over uniformly random row ids: the rate at which the
//                    memory system returns `row_bytes`-wide rows of a table of a given size (L2-,
//                    Infinity-Cache- or HBM-resident), with the access shape the step's gather-type
//                    kernels use (window of n rows per output row, 16-byte lanes, n rows in flight)
//   optimiser stream -- adam_l2's own traffic shape (p, g, m, v read; p, m, v written) over arrays
//                    placed a chosen distance apart inside ONE allocation: isolates what the relative
//                    placement of the four streams does to the rate
#pragma once
#include "common.h"

namespace sert {

__global__ __launch_bounds__(256) void mb_stream_copy(const float4* __restrict__ src, float4* __restrict__ dst,
                                                      size_t n4) {
    const size_t per = (n4 + gridDim.x - 1) / gridDim.x;
    const size_t lo = (size_t)blockIdx.x * per;
    const size_t hi = lo + per < n4 ? lo + per : n4;
    size_t i = lo + threadIdx.x;
    for (; i + 3 * 256 < hi; i += 4 * 256) {   // four loads in flight per lane
        const float4 a = src[i], b = src[i + 256], c = src[i + 512], d = src[i + 768];
        dst[i] = a; dst[i + 256] = b; dst[i + 512] = c; dst[i + 768] = d;
    }
    for (; i < hi; i += 256) dst[i] = src[i];
}

// read-only stream: per-workgroup sums so that nothing is optimised away
__global__ __launch_bounds__(256) void mb_stream_read(const float4* __restrict__ src, size_t n4,
                                                      float* __restrict__ sink) {
    const size_t per = (n4 + gridDim.x - 1) / gridDim.x;
    const size_t lo = (size_t)blockIdx.x * per;
    const size_t hi = lo + per < n4 ? lo + per : n4;
    float s = 0.f;
    size_t i = lo + threadIdx.x;
    for (; i + 3 * 256 < hi; i += 4 * 256) {
        const float4 a = src[i], b = src[i + 256], c = src[i + 512], d = src[i + 768];
        s += (a.x + b.x) + (c.x + d.x) + (a.w + b.w) + (c.w + d.w);
    }
    for (; i < hi; i += 256) s += src[i].x;
    if (s == 1.2345e30f) sink[blockIdx.x] = s;   // (never true for the benchmark's data)
}

// uniformly random row ids in [0, rows): a 32-bit mix of the index
__global__ void mb_fill_ids(uint32_t* __restrict__ ids, size_t count, uint32_t rows, uint32_t salt) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 0x9E3779B9u + salt;
        x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
        ids[i] = (uint32_t)(((uint64_t)x * rows) >> 32);
    }
}

__global__ void mb_fill_f32(float* __restrict__ p, size_t count, float v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x)
        p[i] = v * (float)((i & 1023) + 1) * (1.0f / 1024.0f);
}

}  // namespace sert

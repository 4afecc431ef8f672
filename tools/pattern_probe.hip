// pattern_probe.hip — read-only sweeps of the three candidate load patterns over the
// same 480 MB N x 12 matrix (development tool).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

struct f3 { float x, y, z; };

// (a) coalesced float4, tiles of TILE float4s dealt round-robin
template <int kPerThread>
__global__ void p_float4(const float4* __restrict__ d, int64_t n_vec, uint32_t* sink) {
    uint32_t acc = 0;
    const int64_t tile = (int64_t)blockDim.x * kPerThread;
    for (int64_t t0 = (int64_t)blockIdx.x * tile; t0 < n_vec; t0 += (int64_t)gridDim.x * tile) {
        float4 v[kPerThread];
#pragma unroll
        for (int u = 0; u < kPerThread; ++u) { int64_t i = t0 + u * blockDim.x + threadIdx.x; v[u] = d[i < n_vec ? i : n_vec - 1]; }
#pragma unroll
        for (int u = 0; u < kPerThread; ++u) acc ^= __float_as_uint(v[u].x) ^ __float_as_uint(v[u].y) ^ __float_as_uint(v[u].z) ^ __float_as_uint(v[u].w);
    }
    if (acc == 0x12345678u) sink[blockIdx.x] = acc;
}
// (b) row per lane: 3 x dwordx4 at a 48 B stride
template <int kRows>
__global__ void p_rowlane(const float4* __restrict__ d, int64_t n_rows, uint32_t* sink) {
    uint32_t acc = 0;
    const int64_t tile = (int64_t)blockDim.x * kRows;
    for (int64_t t0 = (int64_t)blockIdx.x * tile; t0 < n_rows; t0 += (int64_t)gridDim.x * tile) {
        float4 v[kRows][3];
#pragma unroll
        for (int u = 0; u < kRows; ++u) { int64_t r = t0 + u * blockDim.x + threadIdx.x; if (r >= n_rows) r = n_rows - 1;
            v[u][0] = d[r * 3]; v[u][1] = d[r * 3 + 1]; v[u][2] = d[r * 3 + 2]; }
#pragma unroll
        for (int u = 0; u < kRows; ++u)
#pragma unroll
            for (int k = 0; k < 3; ++k) acc ^= __float_as_uint(v[u][k].x) ^ __float_as_uint(v[u][k].y) ^ __float_as_uint(v[u][k].z) ^ __float_as_uint(v[u][k].w);
    }
    if (acc == 0x12345678u) sink[blockIdx.x] = acc;
}
// (c) coalesced dwordx3: lane l reads floats [3l, 3l+3) of a 768 B chunk (4 lanes per row)
template <int kPerThread>
__global__ void p_dwordx3(const f3* __restrict__ d, int64_t n_vec, uint32_t* sink) {
    uint32_t acc = 0;
    const int64_t tile = (int64_t)blockDim.x * kPerThread;
    for (int64_t t0 = (int64_t)blockIdx.x * tile; t0 < n_vec; t0 += (int64_t)gridDim.x * tile) {
        f3 v[kPerThread];
#pragma unroll
        for (int u = 0; u < kPerThread; ++u) { int64_t i = t0 + u * blockDim.x + threadIdx.x; v[u] = d[i < n_vec ? i : n_vec - 1]; }
#pragma unroll
        for (int u = 0; u < kPerThread; ++u) acc ^= __float_as_uint(v[u].x) ^ __float_as_uint(v[u].y) ^ __float_as_uint(v[u].z);
    }
    if (acc == 0x12345678u) sink[blockIdx.x] = acc;
}

// (d) fp16 shadow rows: 24 B per row, row per lane as 3 x dwordx2
template <int kRows>
__global__ void p_half_rowlane(const float2* __restrict__ d, int64_t n_rows, uint32_t* sink) {
    uint32_t acc = 0;
    const int64_t tile = (int64_t)blockDim.x * kRows;
    for (int64_t t0 = (int64_t)blockIdx.x * tile; t0 < n_rows; t0 += (int64_t)gridDim.x * tile) {
        float2 v[kRows][3];
#pragma unroll
        for (int u = 0; u < kRows; ++u) { int64_t r = t0 + u * blockDim.x + threadIdx.x; if (r >= n_rows) r = n_rows - 1;
            v[u][0] = d[r * 3]; v[u][1] = d[r * 3 + 1]; v[u][2] = d[r * 3 + 2]; }
#pragma unroll
        for (int u = 0; u < kRows; ++u)
#pragma unroll
            for (int k = 0; k < 3; ++k) acc ^= __float_as_uint(v[u][k].x) ^ __float_as_uint(v[u][k].y);
    }
    if (acc == 0x12345678u) sink[blockIdx.x] = acc;
}

int main() {
    const int64_t n = 10000000;
    float* d; CK(hipMalloc(&d, n * 48)); CK(hipMemset(d, 1, n * 48));
    uint32_t* sink; CK(hipMalloc(&sink, 1 << 20));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto time = [&](auto f) { for (int i = 0; i < 3; ++i) f(); CK(hipDeviceSynchronize()); std::vector<float> ms;
        for (int r = 0; r < 20; ++r) { CK(hipEventRecord(a, 0)); f(); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b)); float t; CK(hipEventElapsedTime(&t, a, b)); ms.push_back(t); }
        std::sort(ms.begin(), ms.end()); return ms[ms.size() / 2] * 1e3f; };
    for (int rep = 0; rep < 2; ++rep)
    for (int block : {256, 512}) for (int bpc : {1, 2, 3, 4}) {
        const int grid = 256 * bpc;
        float f4a = time([&] { hipLaunchKernelGGL(p_float4<3>, dim3(grid), dim3(block), 0, 0, (const float4*)d, n * 3, sink); });
        float f4b = time([&] { hipLaunchKernelGGL(p_float4<6>, dim3(grid), dim3(block), 0, 0, (const float4*)d, n * 3, sink); });
        float r1 = time([&] { hipLaunchKernelGGL(p_rowlane<1>, dim3(grid), dim3(block), 0, 0, (const float4*)d, n, sink); });
        float r2 = time([&] { hipLaunchKernelGGL(p_rowlane<2>, dim3(grid), dim3(block), 0, 0, (const float4*)d, n, sink); });
        float x4 = time([&] { hipLaunchKernelGGL(p_dwordx3<4>, dim3(grid), dim3(block), 0, 0, (const f3*)d, n * 4, sink); });
        float x8 = time([&] { hipLaunchKernelGGL(p_dwordx3<8>, dim3(grid), dim3(block), 0, 0, (const f3*)d, n * 4, sink); });
        float h1 = time([&] { hipLaunchKernelGGL(p_half_rowlane<1>, dim3(grid), dim3(block), 0, 0, (const float2*)d, n, sink); });
        float h2 = time([&] { hipLaunchKernelGGL(p_half_rowlane<2>, dim3(grid), dim3(block), 0, 0, (const float2*)d, n, sink); });
        float h4 = time([&] { hipLaunchKernelGGL(p_half_rowlane<4>, dim3(grid), dim3(block), 0, 0, (const float2*)d, n, sink); });
        printf("   fp16-shadow rows (240 MB): r1 %5.1f r2 %5.1f r4 %5.1f us\n", h1, h2, h4);
        printf("block %3d x %d/CU: float4 u3 %5.1f u6 %5.1f | rowlane r1 %5.1f r2 %5.1f | dwordx3 u4 %5.1f u8 %5.1f  (us)\n", block, bpc, f4a, f4b, r1, r2, x4, x8);
    }
    return 0;
}

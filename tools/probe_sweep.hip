// probe_sweep.hip — how fast can this box stream 480 MB read-only? (development tool)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

template <int kUnroll>
__global__ void probe(const float4* __restrict__ data, int64_t n_vec, uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (kUnroll - 1) * stride < n_vec; i += kUnroll * stride) {
        float4 v[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) v[u] = data[i + u * stride];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) acc ^= __float_as_uint(v[u].x) ^ __float_as_uint(v[u].y) ^ __float_as_uint(v[u].z) ^ __float_as_uint(v[u].w);
    }
    for (; i < n_vec; i += stride) { float4 a = data[i]; acc ^= __float_as_uint(a.x) ^ __float_as_uint(a.w); }
    if (acc == 0x12345678u) sink[blockIdx.x] = acc;
}

// contiguous chunk per block (like the scan kernel)
template <int kUnroll>
__global__ void probe_chunk(const float4* __restrict__ data, int64_t n_vec, uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    const int64_t per = (n_vec + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * per;
    int64_t hi = lo + per; if (hi > n_vec) hi = n_vec;
    int64_t i = lo + threadIdx.x;
    for (; i + (kUnroll - 1) * (int64_t)blockDim.x < hi; i += kUnroll * (int64_t)blockDim.x) {
        float4 v[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) v[u] = data[i + u * (int64_t)blockDim.x];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) acc ^= __float_as_uint(v[u].x) ^ __float_as_uint(v[u].y) ^ __float_as_uint(v[u].z) ^ __float_as_uint(v[u].w);
    }
    for (; i < hi; i += blockDim.x) { float4 a = data[i]; acc ^= __float_as_uint(a.x) ^ __float_as_uint(a.w); }
    if (acc == 0x12345678u) sink[blockIdx.x] = acc;
}

int main() {
    const int64_t n = 10000000; const int64_t n_vec = n * 3;
    float4* d; CK(hipMalloc(&d, n_vec * 16)); CK(hipMemset(d, 1, n_vec * 16));
    uint32_t* sink; CK(hipMalloc(&sink, 1 << 20));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto time = [&](auto f) { for (int i = 0; i < 3; ++i) f(); CK(hipDeviceSynchronize()); std::vector<float> ms;
        for (int r = 0; r < 20; ++r) { CK(hipEventRecord(a, 0)); f(); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b)); float t; CK(hipEventElapsedTime(&t, a, b)); ms.push_back(t); }
        std::sort(ms.begin(), ms.end()); return ms[ms.size() / 2]; };
    for (int block : {256, 512, 1024}) for (int grid : {256, 512, 768, 1024, 2048, 4096}) {
        float t1 = time([&] { hipLaunchKernelGGL(probe<4>, dim3(grid), dim3(block), 0, 0, d, n_vec, sink); });
        float t2 = time([&] { hipLaunchKernelGGL(probe<8>, dim3(grid), dim3(block), 0, 0, d, n_vec, sink); });
        float t3 = time([&] { hipLaunchKernelGGL(probe_chunk<4>, dim3(grid), dim3(block), 0, 0, d, n_vec, sink); });
        float t4 = time([&] { hipLaunchKernelGGL(probe_chunk<8>, dim3(grid), dim3(block), 0, 0, d, n_vec, sink); });
        printf("block %4d grid %4d: strided u4 %6.1f us (%4.0f GB/s)  u8 %6.1f us (%4.0f)  chunk u4 %6.1f us (%4.0f)  u8 %6.1f us (%4.0f)\n", block, grid,
               t1 * 1e3, 0.48 / t1 * 1e3, t2 * 1e3, 0.48 / t2 * 1e3, t3 * 1e3, 0.48 / t3 * 1e3, t4 * 1e3, 0.48 / t4 * 1e3);
    }
    return 0;
}

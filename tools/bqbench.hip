// bqbench.hip — in-process A/B harness for the batched path's pass kernel
// (development tool; not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude \
//         -Ispotify_recommender_amd/csrc tools/bqbench.hip -o tools/bqbench
//   tools/bqbench [rows=12500000] [blocks_per_cu=4]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "replica.hip.h"

using namespace mi355;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

__global__ void fill_kernel(float* p, int64_t n, uint32_t seed) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u ^ seed;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        p[i] = (x >> 8) * (1.0f / 16777216.0f);
    }
}

template <class F> float median_ms(F&& f, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i) f();
    CK(hipDeviceSynchronize());
    std::vector<float> ms;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a, 0)); f(); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float t; CK(hipEventElapsedTime(&t, a, b)); ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
}

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 12500000;
    const int per_cu = argc > 2 ? atoi(argv[2]) : 4;
#ifndef BQ_NB
#define BQ_NB 32   // query blocks (compile with -DBQ_NB=2 for the small-batch kernels)
#endif
    constexpr int NB = BQ_NB;
    float *feats, *queries, *qnorm, *gmax; uint32_t *bfrag, *qflags, *cand_rows, *special; int *cand_count, *counters;
    CK(hipMalloc(&feats, n * 12 * sizeof(float)));
    CK(hipMalloc(&queries, 1024 * 12 * sizeof(float)));
    CK(hipMalloc(&qnorm, 1024 * 4)); CK(hipMalloc(&qflags, 1024 * 4)); CK(hipMalloc(&cand_count, 1024 * 4 * kBqCountStride));
    CK(hipMalloc(&bfrag, NB * 64 * 16)); CK(hipMalloc(&counters, 32)); { int cap = kBqCapMin; CK(hipMemset(counters, 0, 32)); CK(hipMemcpy(counters + 6, &cap, 4, hipMemcpyHostToDevice)); } CK(hipMalloc(&special, kBqSpecialCap * 4));
    CK(hipMalloc(&cand_rows, (size_t)1024 * kBqCapMin * 4));
    const int grid = 256 * per_cu;
    CK(hipMalloc(&gmax, (size_t)grid * NB * 64 * 4));
    fill_kernel<<<4096, 256>>>(feats, n * 12, 1u);
    fill_kernel<<<64, 256>>>(queries, 1024 * 12, 7u);
    bq_prepare_kernel<<<4, 256>>>(queries, 1024, NB, bfrag, qnorm, qflags, cand_count, counters);
    CK(hipDeviceSynchronize());
    const int64_t tiles = (n + 63) / 64;
    const double mfmas_per_simd = double(tiles) * 2 * NB / 1024.0;
    auto report = [&](const char* name, float ms) {
        printf("  %-52s %8.1f us   %6.2f ns per MFMA per SIMD\n", name, ms * 1e3, ms * 1e6 / mfmas_per_simd); fflush(stdout);
    };
    printf("rows %lld, %d workgroups (%d per CU), %d query blocks\n", (long long)n, grid, per_cu, NB);
#define PASS(COLLECT, VAR) [&] { hipLaunchKernelGGL((bq_pass_kernel<NB, COLLECT, VAR>), dim3(grid), dim3(kBqPassBlock), 0, 0, feats, n, tiles, 1, bfrag, gmax, cand_count, cand_rows, counters, special, static_cast<const uint2*>(nullptr)); }
    report("pass 1 (group maxima), product", median_ms(PASS(false, 0), 7));
    report("pass 1, synthetic rows (no HBM reads)", median_ms(PASS(false, 4), 7));
    // pass 2 with thresholds nobody reaches (bfrag threshold slots are 0 -> D = dot >= 0 always hits):
    // set them to -65504 first so that the handler never runs
    std::vector<uint32_t> hb(NB * 64 * 4);
    CK(hipMemcpy(hb.data(), bfrag, hb.size() * 4, hipMemcpyDeviceToHost));
    for (int b = 0; b < NB; ++b) for (int c = 0; c < 32; ++c) hb[(b * 64 + 32 + c) * 4 + 2] = 0x0000fbffu;  // {-65504, 0}
    CK(hipMemcpy(bfrag, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    report("pass 2 (no hits), product", median_ms(PASS(true, 0), 7));
    report("pass 2 (no hits), synthetic rows", median_ms(PASS(true, 4), 7));
    // the same two passes reading the fp16 replica
    uint2* half;
    CK(hipMalloc(&half, (size_t)((n + 1) & ~1ll) * 24));
    replica_build_kernel<<<(unsigned)((((n + 1) & ~1ll) + 255) / 256), 256>>>(feats, n, (n + 1) & ~1ll, half);
    CK(hipDeviceSynchronize());
#define PASSR(COLLECT) [&] { hipLaunchKernelGGL((bq_pass_kernel<NB, COLLECT, 0, true>), dim3(grid), dim3(kBqPassBlock), 0, 0, feats, n, tiles, 1, bfrag, gmax, cand_count, cand_rows, counters, special, static_cast<const uint2*>(half)); }
    report("pass 2 (no hits), fp16 replica", median_ms(PASSR(true), 7));
    report("pass 1 over every tile, fp16 replica", median_ms(PASSR(false), 7));
    // the same kernels on an all-zero catalogue and all-zero fragments (switching power)
    CK(hipMemset(feats, 0, n * 12 * sizeof(float)));
    report("pass 1, all-zero catalogue", median_ms(PASS(false, 0), 7));
    CK(hipMemset(bfrag, 0, NB * 64 * 16));
    report("pass 1, all-zero catalogue and queries", median_ms(PASS(false, 0), 7));
    return 0;
}

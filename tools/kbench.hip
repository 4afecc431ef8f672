// kbench.hip — in-process A/B harness for scan-kernel variants (development
// tool; not part of the product).  hipcc --offload-arch=gfx950 -O3
// -ffp-contract=off -Iinclude -Ispotify_recommender_amd/csrc tools/kbench.hip -o tools/kbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <string>
#include "kernels.hip.h"

using namespace mi355;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

// ---- variant: row-per-lane loads only (no math) ----
__global__ __launch_bounds__(kBlock) void load_only_rowlane(const float* __restrict__ feats, int64_t n, int64_t rows_per_block, int iters, uint32_t* sink) {
    const int tid = threadIdx.x;
    const int64_t blk_begin = (int64_t)blockIdx.x * rows_per_block;
    const int64_t last_row = n - 1;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        const int64_t tile_begin = blk_begin + (int64_t)it * kTileRows;
        Row r[kRowsPerThread];
#pragma unroll
        for (int u = 0; u < kRowsPerThread; ++u) {
            int64_t row = tile_begin + u * kBlock + tid;
            r[u] = load_row(feats, row < last_row ? row : last_row);
        }
#pragma unroll
        for (int u = 0; u < kRowsPerThread; ++u) {
            acc ^= __float_as_uint(r[u].a.x) ^ __float_as_uint(r[u].a.w) ^ __float_as_uint(r[u].b.y) ^ __float_as_uint(r[u].c.z)
                 ^ __float_as_uint(r[u].a.y) ^ __float_as_uint(r[u].a.z) ^ __float_as_uint(r[u].b.x) ^ __float_as_uint(r[u].b.z)
                 ^ __float_as_uint(r[u].b.w) ^ __float_as_uint(r[u].c.x) ^ __float_as_uint(r[u].c.y) ^ __float_as_uint(r[u].c.w);
        }
    }
    if (acc == 0x12345678u) sink[blockIdx.x] = acc;
}

// ---- variant: coalesced loads + LDS transpose, scores only ----
template <int U, bool kMath>
__global__ __launch_bounds__(kBlock) void lds_transpose_scores(const float* __restrict__ feats, int64_t n, int64_t rows_per_block, int iters,
                                                               QueryArg qarg, float* __restrict__ out, uint32_t* sink) {
    __shared__ float4 s_stage[kBlock / 64][192];  // 3 KiB per wave
    float q[kDim];
#pragma unroll
    for (int j = 0; j < kDim; ++j) q[j] = qarg.q[j];
    const float qn = query_norm(q);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t blk_begin = (int64_t)blockIdx.x * rows_per_block;
    int64_t blk_end = blk_begin + rows_per_block; if (blk_end > n) blk_end = n;
    const int64_t max_f4 = n * 3 - 1;
    const float4* f4 = reinterpret_cast<const float4*>(feats);
    float4* stage = s_stage[wave];
    uint32_t acc = 0;
    constexpr int kTile = kBlock * U;

    float4 cur[U][3], nxt[U][3];
    auto load_tile = [&](float4 (&dst)[U][3], int it) {
        const int64_t wave_row0 = blk_begin + (int64_t)it * kTile + (int64_t)wave * (64 * U);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                int64_t idx = (wave_row0 + u * 64) * 3 + k * 64 + lane;
                dst[u][k] = f4[idx < max_f4 ? idx : max_f4];
            }
    };
    auto process = [&](float4 (&src)[U][3], int it) {
        const int64_t wave_row0 = blk_begin + (int64_t)it * kTile + (int64_t)wave * (64 * U);
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int k = 0; k < 3; ++k) stage[k * 64 + lane] = src[u][k];
            Row r;
            r.a = stage[lane * 3 + 0];
            r.b = stage[lane * 3 + 1];
            r.c = stage[lane * 3 + 2];
            const int64_t row = wave_row0 + u * 64 + lane;
            if (kMath) {
                const float s = cosine_score(q, qn, r);
                if (row < blk_end) out[row] = s;
            } else {
                acc ^= __float_as_uint(r.a.x) ^ __float_as_uint(r.a.w) ^ __float_as_uint(r.b.y) ^ __float_as_uint(r.c.z)
                     ^ __float_as_uint(r.a.y) ^ __float_as_uint(r.a.z) ^ __float_as_uint(r.b.x) ^ __float_as_uint(r.b.z)
                     ^ __float_as_uint(r.b.w) ^ __float_as_uint(r.c.x) ^ __float_as_uint(r.c.y) ^ __float_as_uint(r.c.w);
            }
        }
    };
    load_tile(cur, 0);
    for (int it = 0; it < iters; it += 2) {
        if (it + 1 < iters) load_tile(nxt, it + 1);
        process(cur, it);
        if (it + 1 < iters) {
            if (it + 2 < iters) load_tile(cur, it + 2);
            process(nxt, it + 1);
        }
    }
    if (!kMath && acc == 0x12345678u) sink[blockIdx.x] = acc;
}

struct Timer {
    hipEvent_t a, b;
    Timer() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
    template <class F> float run(F&& f, int reps) {
        for (int i = 0; i < 3; ++i) f();
        CK(hipDeviceSynchronize());
        std::vector<float> ms;
        for (int r = 0; r < reps; ++r) {
            CK(hipEventRecord(a, 0)); f(); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
            float t; CK(hipEventElapsedTime(&t, a, b)); ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        return ms[ms.size() / 2];
    }
};

int main(int argc, char** argv) {
    int64_t n = argc > 1 ? atoll(argv[1]) : 10000000;
    int reps = argc > 2 ? atoi(argv[2]) : 20;
    float* d_feats; CK(hipMalloc(&d_feats, n * 48));
    std::vector<float> h(n * 12);
    uint64_t s = 88172645463325252ull;
    for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (s >> 40) * (1.0f / 16777216.0f); }
    CK(hipMemcpy(d_feats, h.data(), n * 48, hipMemcpyHostToDevice));
    float* d_scores; CK(hipMalloc(&d_scores, n * 4));
    float* d_scores2; CK(hipMalloc(&d_scores2, n * 4));
    uint32_t* d_sink; CK(hipMalloc(&d_sink, 1 << 20));
    uint64_t* d_lists; CK(hipMalloc(&d_lists, 8ull * 2048 * 1024));
    uint64_t* d_out; CK(hipMalloc(&d_out, 8 * 1024));
    QueryArg qa; for (int j = 0; j < 12; ++j) qa.q[j] = h[12 * 7919 + j];
    Timer T;
    const double gb = n * 48.0 / 1e9;
    auto report = [&](const char* name, float ms) { printf("%-44s %8.3f us  %8.1f GB/s\n", name, ms * 1e3, gb / (ms * 1e-3)); };

    for (int bpc : {1, 2}) {
        int64_t maxb = 256 * bpc; int64_t rpb = (n + maxb - 1) / maxb; rpb = (rpb + 63) / 64 * 64;
        int grid = (int)((n + rpb - 1) / rpb); int iters = (int)((rpb + kTileRows - 1) / kTileRows);
        printf("-- blocks/CU %d grid %d rows/block %lld iters %d\n", bpc, grid, (long long)rpb, iters);
        report("stream_probe (coalesced f4)", T.run([&] { hipLaunchKernelGGL(stream_probe_kernel, dim3(grid), dim3(kBlock), 0, 0, (const float4*)d_feats, n * 3, d_sink); }, reps));
        report("load_only_rowlane", T.run([&] { hipLaunchKernelGGL(load_only_rowlane, dim3(grid), dim3(kBlock), 0, 0, d_feats, n, rpb, iters, d_sink); }, reps));
        report("scan scores-only (rowlane)", T.run([&] { hipLaunchKernelGGL((scan_kernel<false, true>), dim3(grid), dim3(kBlock), 0, 0, d_feats, n, rpb, iters, (int64_t)0, qa, (int64_t)0, (int64_t)-1, 1, (uint64_t*)nullptr, d_scores); }, reps));
        report("scan topk=100 (rowlane)", T.run([&] { hipLaunchKernelGGL((scan_kernel<false, false>), dim3(grid), dim3(kBlock), 0, 0, d_feats, n, rpb, iters, (int64_t)0, qa, (int64_t)0, (int64_t)7919, 100, d_lists, (float*)nullptr); }, reps));
        report("  dbg1 no tile barriers", T.run([&] { hipLaunchKernelGGL((scan_kernel<false, false, 1>), dim3(grid), dim3(kBlock), 0, 0, d_feats, n, rpb, iters, (int64_t)0, qa, (int64_t)0, (int64_t)7919, 100, d_lists, (float*)nullptr); }, reps));
        report("  dbg4 preset thr (no seed)", T.run([&] { hipLaunchKernelGGL((scan_kernel<false, false, 4>), dim3(grid), dim3(kBlock), 0, 0, d_feats, n, rpb, iters, (int64_t)0, qa, (int64_t)0, (int64_t)7919, 100, d_lists, (float*)nullptr); }, reps));
        report("  dbg5 preset thr, no barriers", T.run([&] { hipLaunchKernelGGL((scan_kernel<false, false, 5>), dim3(grid), dim3(kBlock), 0, 0, d_feats, n, rpb, iters, (int64_t)0, qa, (int64_t)0, (int64_t)7919, 100, d_lists, (float*)nullptr); }, reps));
        report("  dbg7 preset thr, no barriers, no append", T.run([&] { hipLaunchKernelGGL((scan_kernel<false, false, 7>), dim3(grid), dim3(kBlock), 0, 0, d_feats, n, rpb, iters, (int64_t)0, qa, (int64_t)0, (int64_t)7919, 100, d_lists, (float*)nullptr); }, reps));
        report("merge topk=100", T.run([&] { hipLaunchKernelGGL(merge_kernel, dim3(1), dim3(kMergeBlock), 0, 0, d_lists, grid, 100, (int64_t)0, 100, d_out, (int64_t*)nullptr, (float*)nullptr, (int64_t)0); }, reps));
    }
    // correctness of the transposed variant vs the rowlane one
    std::vector<float> a(n), b(n);
    CK(hipMemcpy(a.data(), d_scores, n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), d_scores2, n * 4, hipMemcpyDeviceToHost));
    int64_t bad = 0; for (int64_t i = 0; i < n; ++i) bad += (a[i] != b[i]);
    printf("transposed vs rowlane score mismatches: %lld\n", (long long)bad);
    return 0;
}

// kbench.hip — in-process A/B harness for scan-kernel geometries (development
// tool; not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude \
//         -Ispotify_recommender_amd/csrc tools/kbench.hip -o tools/kbench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "kernels.hip.h"

using namespace mi355;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

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

static int64_t g_n; static float* d_feats; static float* d_scores; static uint64_t* d_lists; static uint64_t* d_out; static uint32_t* d_sink;
static QueryArg g_qa; static int g_reps; static int g_topk; static double g_gb;
static std::vector<uint64_t> g_ref; static int g_stages = 0;

static void report(const char* name, float ms) { printf("  %-40s %8.3f us  %8.1f GB/s\n", name, ms * 1e3, g_gb / (ms * 1e-3)); fflush(stdout); }

template <typename Cfg>
void run_cfg(const char* label, int blocks_per_cu) {
    Timer T;
    int64_t maxb = 256 * blocks_per_cu; if (maxb > kMergeMaxLists) maxb = kMergeMaxLists;
    int64_t rpb = (g_n + maxb - 1) / maxb; rpb = (rpb + 63) / 64 * 64;
    int grid = (int)((g_n + rpb - 1) / rpb); int iters = (int)((rpb + Cfg::kTileRows - 1) / Cfg::kTileRows);
    int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, scan_kernel<Cfg, false, false>, Cfg::kBlock, 0));
    printf("-- %s: block %d x %d rows, depth %d, minwaves %d, grid %d (%d/CU asked, occupancy API %d), rows/block %lld, iters %d\n", label, Cfg::kBlock, Cfg::kRowsPerThread, Cfg::kDepth, Cfg::kMinWaves, grid, blocks_per_cu, occ, (long long)rpb, iters);
    report("scores-only", T.run([&] { hipLaunchKernelGGL((scan_kernel<Cfg, false, true>), dim3(grid), dim3(Cfg::kBlock), 0, 0, d_feats, g_n, rpb, iters, (int64_t)0, g_qa, (int64_t)0, (int64_t)-1, 1, (uint64_t*)nullptr, d_scores, (const uint64_t*)nullptr, PrevMerge{nullptr, 0, 0, nullptr}); }, g_reps));
    report("topk", T.run([&] { hipLaunchKernelGGL((scan_kernel<Cfg, false, false>), dim3(grid), dim3(Cfg::kBlock), 0, 0, d_feats, g_n, rpb, iters, (int64_t)0, g_qa, (int64_t)0, (int64_t)7919, g_topk, d_lists, (float*)nullptr, (const uint64_t*)nullptr, PrevMerge{nullptr, 0, 0, nullptr}); }, g_reps));
    float mm = T.run([&] { hipLaunchKernelGGL(merge_kernel, dim3(1), dim3(kMergeBlock), 0, 0, d_lists, grid, g_topk, (int64_t)g_topk, (int64_t)0, g_topk, d_out, (int64_t*)nullptr, (float*)nullptr, (int64_t)0); }, g_reps);
    report("merge", mm);
    report("  empty-ish kernel (probe, 64 vec)", T.run([&] { hipLaunchKernelGGL(stream_probe_kernel, dim3(1), dim3(kProbeBlock), 0, 0, (const float4*)d_feats, (int64_t)64, d_sink); }, g_reps));
    std::vector<uint64_t> got(g_topk);
    CK(hipMemcpy(got.data(), d_out, 8 * g_topk, hipMemcpyDeviceToHost));
    if (g_ref.empty()) g_ref = got;
    printf("  result %s\n", got == g_ref ? "matches first config" : "DIFFERS from first config");
    {
        int64_t tiles = (g_n + Cfg::kTileRows - 1) / Cfg::kTileRows;
        int ig = (int)std::min<int64_t>(256 * blocks_per_cu, tiles); int it2 = (int)((tiles + ig - 1) / ig);
        report("topk, interleaved tiles", T.run([&] { hipLaunchKernelGGL((scan_kernel<Cfg, false, false>), dim3(ig), dim3(Cfg::kBlock), 0, 0, d_feats, g_n, (int64_t)0, it2, (int64_t)0, g_qa, (int64_t)0, (int64_t)7919, g_topk, d_lists, (float*)nullptr, (const uint64_t*)nullptr, PrevMerge{nullptr, 0, 0, nullptr}); }, g_reps));
        hipLaunchKernelGGL(merge_kernel, dim3(1), dim3(kMergeBlock), 0, 0, d_lists, ig, g_topk, (int64_t)g_topk, (int64_t)0, g_topk, d_out, (int64_t*)nullptr, (float*)nullptr, (int64_t)0);
        std::vector<uint64_t> got2(g_topk); CK(hipMemcpy(got2.data(), d_out, 8 * g_topk, hipMemcpyDeviceToHost));
        printf("  interleaved result %s\n", got2 == g_ref ? "matches" : "DIFFERS");
        report("topk contiguous (again)", T.run([&] { hipLaunchKernelGGL((scan_kernel<Cfg, false, false>), dim3(grid), dim3(Cfg::kBlock), 0, 0, d_feats, g_n, rpb, iters, (int64_t)0, g_qa, (int64_t)0, (int64_t)7919, g_topk, d_lists, (float*)nullptr, (const uint64_t*)nullptr, PrevMerge{nullptr, 0, 0, nullptr}); }, g_reps));
        report("topk, interleaved tiles (again)", T.run([&] { hipLaunchKernelGGL((scan_kernel<Cfg, false, false>), dim3(ig), dim3(Cfg::kBlock), 0, 0, d_feats, g_n, (int64_t)0, it2, (int64_t)0, g_qa, (int64_t)0, (int64_t)7919, g_topk, d_lists, (float*)nullptr, (const uint64_t*)nullptr, PrevMerge{nullptr, 0, 0, nullptr}); }, g_reps));
    }
    report("topk, preset threshold (floor)", T.run([&] { hipLaunchKernelGGL((scan_kernel<Cfg, false, false, 4>), dim3(grid), dim3(Cfg::kBlock), 0, 0, d_feats, g_n, rpb, iters, (int64_t)0, g_qa, (int64_t)0, (int64_t)7919, g_topk, d_lists, (float*)nullptr, (const uint64_t*)nullptr, PrevMerge{nullptr, 0, 0, nullptr}); }, g_reps));
}

template <typename Cfg>
void run_multi(const char* label, int blocks_per_cu, int nq, const std::vector<float>& h) {
    Timer T;
    int64_t maxb = 256 * blocks_per_cu; if (maxb > kMergeMaxLists) maxb = kMergeMaxLists;
    int64_t rpb = (g_n + maxb - 1) / maxb; rpb = (rpb + 63) / 64 * 64;
    int grid = (int)((g_n + rpb - 1) / rpb); int iters = (int)((rpb + Cfg::kTileRows - 1) / Cfg::kTileRows);
    int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, scan_multi_kernel<Cfg>, Cfg::kBlock, 0));
    MultiQueryArg qa; memset(&qa, 0, sizeof qa);
    for (int q = 0; q < kMultiQueries; ++q) { qa.exclude[q] = 7919 + 1000 * q; for (int j = 0; j < 12; ++j) qa.q[q][j] = h[12 * (7919 + 1000 * q) + j]; }
    printf("-- multi %s: block %d x %d rows, minwaves %d, grid %d (%d/CU asked, occupancy API %d), queries %d\n", label, Cfg::kBlock, Cfg::kRowsPerThread, Cfg::kMinWaves, grid, blocks_per_cu, occ, nq);
    float ms = T.run([&] { hipLaunchKernelGGL((scan_multi_kernel<Cfg>), dim3(grid), dim3(Cfg::kBlock), 0, 0, d_feats, g_n, rpb, rpb, iters, (int64_t)0, qa, nq, 0, g_topk, d_lists, (const uint64_t*)nullptr, PrevMerge{nullptr, 0, 0, nullptr}); }, g_reps);
    printf("  %-40s %8.3f us  %8.1f GB/s  %9.0f queries/s\n", "multi-query pass (no seed)", ms * 1e3, g_gb / (ms * 1e-3), nq / (ms * 1e-3));
    static uint64_t* d_seed = nullptr; if (!d_seed) CK(hipMalloc(&d_seed, 8 * 8 * 1024));
    float ms1 = T.run([&] { hipLaunchKernelGGL((scan_multi_kernel<Cfg>), dim3(grid), dim3(Cfg::kBlock), 0, 0, d_feats, g_n, (int64_t)Cfg::kTileRows, rpb, 1, (int64_t)0, qa, nq, 0, g_topk, d_lists, (const uint64_t*)nullptr, PrevMerge{nullptr, 0, 0, nullptr}); }, g_reps);
    report("seed pass (first tile of every workgroup)", ms1);
    hipLaunchKernelGGL(merge_kernel, dim3(nq), dim3(kMergeBlock), 0, 0, d_lists, grid, g_topk, (int64_t)g_topk, (int64_t)grid * g_topk, g_topk, d_seed, (int64_t*)nullptr, (float*)nullptr, (int64_t)g_topk);
    CK(hipDeviceSynchronize());
    ms = T.run([&] { hipLaunchKernelGGL((scan_multi_kernel<Cfg>), dim3(grid), dim3(Cfg::kBlock), 0, 0, d_feats, g_n, rpb, rpb, iters, (int64_t)0, qa, nq, 0, g_topk, d_lists, (const uint64_t*)d_seed); }, g_reps);
    printf("  %-40s %8.3f us  %8.1f GB/s  %9.0f queries/s\n", "multi-query pass (seeded)", ms * 1e3, g_gb / (ms * 1e-3), nq / (ms * 1e-3));
    float chain = T.run([&] {
        hipLaunchKernelGGL((scan_multi_kernel<Cfg>), dim3(grid), dim3(Cfg::kBlock), 0, 0, d_feats, g_n, (int64_t)Cfg::kTileRows, rpb, 1, (int64_t)0, qa, nq, 0, g_topk, d_lists, (const uint64_t*)nullptr);
        hipLaunchKernelGGL(merge_kernel, dim3(nq), dim3(kMergeBlock), 0, 0, d_lists, grid, g_topk, (int64_t)g_topk, (int64_t)grid * g_topk, g_topk, d_seed, (int64_t*)nullptr, (float*)nullptr, (int64_t)g_topk);
        hipLaunchKernelGGL((scan_multi_kernel<Cfg>), dim3(grid), dim3(Cfg::kBlock), 0, 0, d_feats, g_n, rpb, rpb, iters, (int64_t)0, qa, nq, 0, g_topk, d_lists, (const uint64_t*)d_seed);
        hipLaunchKernelGGL(merge_kernel, dim3(nq), dim3(kMergeBlock), 0, 0, d_lists, grid, g_topk, (int64_t)g_topk, (int64_t)grid * g_topk, g_topk, d_out, (int64_t*)nullptr, (float*)nullptr, (int64_t)g_topk);
    }, g_reps);
    printf("  %-40s %8.3f us  %8.1f GB/s  %9.0f queries/s\n", "seed+merge+pass+merge chain", chain * 1e3, g_gb / (chain * 1e-3), nq / (chain * 1e-3));
    float mm = T.run([&] { hipLaunchKernelGGL(merge_kernel, dim3(nq), dim3(kMergeBlock), 0, 0, d_lists, grid, g_topk, (int64_t)g_topk, (int64_t)grid * g_topk, g_topk, d_out, (int64_t*)nullptr, (float*)nullptr, (int64_t)g_topk); }, g_reps);
    report("merge (one workgroup per query)", mm);
    std::vector<uint64_t> got(g_topk);
    CK(hipMemcpy(got.data(), d_out, 8 * g_topk, hipMemcpyDeviceToHost));
    if (!g_ref.empty()) printf("  query 0 result %s\n", got == g_ref ? "matches the single-query kernel" : "DIFFERS from the single-query kernel");
}

int main(int argc, char** argv) {
    g_n = argc > 1 ? atoll(argv[1]) : 10000000;
    g_reps = argc > 2 ? atoi(argv[2]) : 20;
    g_topk = argc > 3 ? atoi(argv[3]) : 100;
    g_gb = g_n * 48.0 / 1e9;
    g_stages = 0;
    CK(hipMalloc(&d_feats, g_n * 48));
    std::vector<float> h(g_n * 12);
    uint64_t s = 88172645463325252ull;
    for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (s >> 40) * (1.0f / 16777216.0f); }
    CK(hipMemcpy(d_feats, h.data(), g_n * 48, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_scores, g_n * 4));
    CK(hipMalloc(&d_sink, 1 << 20));
    CK(hipMalloc(&d_lists, 8ull * 2048 * 1024));
    CK(hipMalloc(&d_out, 8 * 1024 * 16));
    for (int j = 0; j < 12; ++j) g_qa.q[j] = h[12 * 7919 + j];
    {
        Timer T;
        printf("-- stream probe (coalesced float4 reads)\n");
        for (int g : {256, 512, 768}) { char nm[64]; snprintf(nm, sizeof nm, "grid %d", g);
            report(nm, T.run([&] { hipLaunchKernelGGL(stream_probe_kernel, dim3(g), dim3(kProbeBlock), 0, 0, (const float4*)d_feats, g_n * 3, d_sink); }, g_reps)); }
    }
    const std::string mode = argc > 4 ? argv[4] : "default";
    if (mode == "merge8") {   // the per-rank merge of an 8-GPU run: 8 sorted lists of topk keys
        Timer T;
        std::vector<uint64_t> keys(8 * g_topk);
        uint64_t sd = 12345;
        for (int l = 0; l < 8; ++l) { std::vector<uint64_t> v(g_topk); for (auto& k : v) { sd ^= sd << 13; sd ^= sd >> 7; sd ^= sd << 17; k = (0xBF700000ull + (sd & 0xFFFFF)) << 32 | (sd >> 40); }
            std::sort(v.rbegin(), v.rend()); std::copy(v.begin(), v.end(), keys.begin() + l * g_topk); }
        CK(hipMemcpy(d_lists, keys.data(), keys.size() * 8, hipMemcpyHostToDevice));
        report("merge of 8 rank lists", T.run([&] { hipLaunchKernelGGL(merge_kernel, dim3(1), dim3(kMergeBlock), 0, 0, d_lists, 8, g_topk, (int64_t)g_topk, (int64_t)0, g_topk, d_out, (int64_t*)nullptr, (float*)nullptr, (int64_t)0); }, g_reps));
        std::vector<uint64_t> got(g_topk); CK(hipMemcpy(got.data(), d_out, 8 * g_topk, hipMemcpyDeviceToHost));
        std::sort(keys.rbegin(), keys.rend());
        printf("  result %s\n", std::equal(got.begin(), got.end(), keys.begin()) ? "correct" : "WRONG");
        report("empty-ish kernel", T.run([&] { hipLaunchKernelGGL(stream_probe_kernel, dim3(1), dim3(kProbeBlock), 0, 0, (const float4*)d_feats, (int64_t)64, d_sink); }, g_reps));
        return 0;
    }
    if (mode == "multi8") {   // one clean configuration for counter collection
        const int nq = argc > 5 ? atoi(argv[5]) : 8;   // <= kMultiQueries of this build
        run_multi<MultiCfg<512, 1, 4>>("M512x1", 2, nq, h);
        return 0;
    }
    if (mode == "sweep") {
    run_cfg<ScanCfg<512, 1, 6, 2>>("I d2", 3);
    run_cfg<ScanCfg<512, 1, 6, 2>>("I d2", 1);
    run_cfg<ScanCfg<512, 1, 2, 1>>("I d1", 1);
    run_cfg<ScanCfg<512, 1, 2, 1>>("I d1", 2);
    run_cfg<ScanCfg<512, 2, 2, 1>>("B d1", 1);
    run_cfg<ScanCfg<512, 3, 2, 1>>("F d1", 1);
    run_cfg<ScanCfg<1024, 1, 2, 1>>("L d1", 1);
    run_cfg<ScanCfg<1024, 1, 2, 2>>("L d2", 1);
    run_cfg<ScanCfg<256, 2, 2, 1>>("D d1", 1);
    run_cfg<ScanCfg<256, 2, 2, 1>>("D d1", 2);
    run_cfg<ScanCfg<256, 1, 2, 2>>("K d2", 1);
    run_cfg<ScanCfg<256, 1, 2, 2>>("K d2", 2);
    } else {
    run_cfg<ScanCfg<512, 1, 6>>("I", 3);
    if (mode == "multi") {
    run_multi<MultiCfg<512, 1, 4>>("M512x1", 2, 8, h);
    run_multi<MultiCfg<512, 1, 4>>("M512x1", 2, 4, h);
    run_multi<MultiCfg<512, 1, 4>>("M512x1", 2, 1, h);
    run_multi<MultiCfg<256, 2, 4>>("M256x2", 3, 8, h);
    run_multi<MultiCfg<256, 1, 4>>("M256x1", 4, 8, h);
    run_multi<MultiCfg<1024, 1, 4>>("M1024x1", 1, 8, h);
    }
    }
    return 0;
}

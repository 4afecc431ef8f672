// How fast are same-address device-scope atomics from many workgroups on gfx950?  (Decides whether the
// single-query scan can hand out tiles dynamically.)   hipcc --offload-arch=gfx950 -O3 -o atomic_probe atomic_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(unsigned* counter, int per_wg, unsigned long long* lat, int spread) {
    if (threadIdx.x != 0) return;
    unsigned* c = counter + (spread ? (blockIdx.x % spread) * 64 : 0);
    unsigned long long t0 = wall_clock64();
    unsigned acc = 0;
    for (int i = 0; i < per_wg; ++i) acc += atomicAdd(c, 1u);
    unsigned long long t1 = wall_clock64();
    lat[blockIdx.x] = t1 - t0 + (acc == 0xffffffffu);
}

int main() {
    unsigned* counter;
    unsigned long long* lat;
    (void)hipMalloc(&counter, 64 * 64 * 4);
    (void)hipMalloc(&lat, 1024 * 8);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    for (int spread : {0, 8}) {
        for (int wgs : {1, 64, 512}) {
            for (int per : {1, 10, 100}) {
                (void)hipMemset(counter, 0, 64 * 64 * 4);
                probe<<<wgs, 64>>>(counter, per, lat, spread);   // warm
                (void)hipDeviceSynchronize();
                (void)hipEventRecord(a);
                probe<<<wgs, 64>>>(counter, per, lat, spread);
                (void)hipEventRecord(b);
                (void)hipEventSynchronize(b);
                float ms;
                (void)hipEventElapsedTime(&ms, a, b);
                std::vector<unsigned long long> h(wgs);
                (void)hipMemcpy(h.data(), lat, wgs * 8, hipMemcpyDeviceToHost);
                unsigned long long mx = 0, sum = 0;
                for (auto v : h) { mx = v > mx ? v : mx; sum += v; }
                printf("{\"counters\": %d, \"wgs\": %d, \"per_wg\": %d, \"kernel_us\": %.2f, \"wg_loop_us_mean\": %.2f, \"wg_loop_us_max\": %.2f, \"ns_per_atomic_serialised\": %.1f}\n",
                       spread ? spread : 1, wgs, per, ms * 1e3, sum / 100.0 / wgs, mx / 100.0, mx * 10.0 / (double(wgs) * per / (spread ? spread : 1)));
            }
        }
    }
    return 0;
}

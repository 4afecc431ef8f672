// Operand layout of v_mfma_i32_32x32x32_i8 on gfx950, checked against a CPU product:
//   A: lane l holds row m = l % 32, k = 16 (l / 32) + 4 w + b for byte b of dword w
//   B: lane l holds column n = l % 32, same k
//   D: lane l holds column n = l % 32; register v is row m = 8 (v / 4) + 4 (l / 32) + (v % 4)
// hipcc --offload-arch=gfx950 -O3 -o mfma_i8_layout mfma_i8_layout.hip && ./mfma_i8_layout
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
__global__ void k(const v4i* a, const v4i* b, v16i* d) {
    v16i z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    d[threadIdx.x] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[threadIdx.x], b[threadIdx.x], z, 0, 0, 0);
}
int main() {
    int8_t A[32][32], B[32][32];
    srand(7);
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            A[i][j] = static_cast<int8_t>(rand() % 255 - 127);
            B[i][j] = static_cast<int8_t>(rand() % 255 - 127);
        }
    int8_t ha[64][16], hb[64][16];
    for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 16; ++e) {
            ha[l][e] = A[l % 32][16 * (l / 32) + e];
            hb[l][e] = B[16 * (l / 32) + e][l % 32];
        }
    void *da, *db, *dd;
    (void)hipMalloc(&da, sizeof ha);
    (void)hipMalloc(&db, sizeof hb);
    (void)hipMalloc(&dd, 64 * 64);
    (void)hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice);
    (void)hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
    k<<<1, 64>>>(static_cast<v4i*>(da), static_cast<v4i*>(db), static_cast<v16i*>(dd));
    int hd[64][16];
    (void)hipMemcpy(hd, dd, sizeof hd, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int v = 0; v < 16; ++v) {
            const int m = 8 * (v / 4) + 4 * (l / 32) + (v % 4), n = l % 32;
            int want = 0;
            for (int kk = 0; kk < 32; ++kk) want += static_cast<int>(A[m][kk]) * static_cast<int>(B[kk][n]);
            bad += hd[l][v] != want;
        }
    printf("{\"mfma_i32_32x32x32_i8_layout_mismatches\": %d}\n", bad);
    return bad != 0;
}

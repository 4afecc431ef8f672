// mfma_probe.hip — how many VALU reductions fit beside one v_mfma_f32_32x32x16_f16 per
// SIMD, by waves per SIMD, operand count and dependency shape (development probe for
// csrc/batched.hip.h; not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_probe tools/mfma_probe.hip && tools/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

// MODE 0: K x v_max3_i32 in ONE dependent chain on the MFMA result of the previous iteration
// MODE 1: the same K ops as a tree (depth 3)
// MODE 2: K x v_max3_i32 on 4 independent accumulators, operands NOT from the MFMA
// MODE 3: K x v_max_i32 (2 operands), 4 independent accumulators, operands NOT from the MFMA
// MODE 4: no VALU
template <int MODE, int K>
__global__ __launch_bounds__(256) void probe(const uint4* in, int* out, int iters) {
    h8 A = __builtin_bit_cast(h8, in[threadIdx.x & 63]);
    h8 B = __builtin_bit_cast(h8, in[64 + (threadIdx.x & 63)]);
    const f16v zero = {0};
    int m0 = 0, m1 = 0, m2 = 0, m3 = 0;
    int x0 = threadIdx.x, x1 = threadIdx.x * 3, x2 = threadIdx.x * 5, x3 = threadIdx.x * 7;
    f16v D0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, zero, 0, 0, 0);
    for (int it = 0; it < iters; ++it) {
        f16v D1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, zero, 0, 0, 0);
        if constexpr (MODE == 0) {
#pragma unroll
            for (int k = 0; k < K; ++k)
                asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(m0) : "v"(D0[(2 * k) & 15]), "v"(D0[(2 * k + 1) & 15]));
        } else if constexpr (MODE == 1) {
            int t0, t1, t2, t3, t4;
            asm volatile("v_max3_i32 %0, %1, %2, %3" : "=v"(t0) : "v"(D0[0]), "v"(D0[1]), "v"(D0[2]));
            asm volatile("v_max3_i32 %0, %1, %2, %3" : "=v"(t1) : "v"(D0[3]), "v"(D0[4]), "v"(D0[5]));
            asm volatile("v_max3_i32 %0, %1, %2, %3" : "=v"(t2) : "v"(D0[6]), "v"(D0[7]), "v"(D0[8]));
            asm volatile("v_max3_i32 %0, %1, %2, %3" : "=v"(t3) : "v"(D0[9]), "v"(D0[10]), "v"(D0[11]));
            asm volatile("v_max3_i32 %0, %1, %2, %3" : "=v"(t4) : "v"(D0[12]), "v"(D0[13]), "v"(D0[14]));
            asm volatile("v_max3_i32 %0, %1, %2, %3" : "=v"(t0) : "v"(t0), "v"(t1), "v"(t2));
            asm volatile("v_max3_i32 %0, %1, %2, %3" : "=v"(t3) : "v"(t3), "v"(t4), "v"(D0[15]));
            asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(m0) : "v"(t0), "v"(t3));
        } else if constexpr (MODE == 2) {
#pragma unroll
            for (int k = 0; k < K; k += 4) {
                asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(m0) : "v"(x0), "v"(x1));
                asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(m1) : "v"(x1), "v"(x2));
                asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(m2) : "v"(x2), "v"(x3));
                asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(m3) : "v"(x3), "v"(x0));
            }
        } else if constexpr (MODE == 3) {
#pragma unroll
            for (int k = 0; k < K; k += 4) {
                asm volatile("v_max_i32 %0, %0, %1" : "+v"(m0) : "v"(x0));
                asm volatile("v_max_i32 %0, %0, %1" : "+v"(m1) : "v"(x1));
                asm volatile("v_max_i32 %0, %0, %1" : "+v"(m2) : "v"(x2));
                asm volatile("v_max_i32 %0, %0, %1" : "+v"(m3) : "v"(x3));
            }
        }
        D0 = D1;
    }
    out[blockIdx.x * 256 + threadIdx.x] = m0 + m1 + m2 + m3 + __float_as_int(D0[0]);
}

// MODE 5: the pass kernel's shape: two accumulator tiles alternate, the 8-op tree
//         reduces the tile of the PREVIOUS MFMA while the next one runs
// MODE 6: same instruction stream, but the tree reads registers no MFMA writes
// MODE 7: MODE 5 + one ds_read_b128 per MFMA feeding the B operand
template <int MODE>
__global__ __launch_bounds__(256) void probe2(const uint4* in, int* out, int iters) {
    __shared__ uint4 s_b[32][64];
    for (int i = threadIdx.x; i < 32 * 64; i += 256) (&s_b[0][0])[i] = in[i & 127];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    h8 A = __builtin_bit_cast(h8, in[lane]);
    h8 B = __builtin_bit_cast(h8, in[64 + lane]);
    const f16v zero = {0};
    int m = 0;
    f16v X, Y;
    for (int i = 0; i < 16; ++i) { X[i] = threadIdx.x + i; Y[i] = threadIdx.x * 3 + i; }
    auto tree = [&](const f16v& d) {
        auto b = [&](int i) { return __float_as_int(d[i]); };
        auto max3 = [](int x, int y, int z) { return max(max(x, y), z); };
        const int t0 = max3(b(0), b(1), b(2)), t1 = max3(b(3), b(4), b(5)), t2 = max3(b(6), b(7), b(8));
        const int t3 = max3(b(9), b(10), b(11)), t4 = max3(b(12), b(13), b(14));
        m = max3(m, max3(t0, t1, t2), max3(t3, t4, b(15)));
    };
    f16v D0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, zero, 0, 0, 0);
    f16v D1;
    for (int it = 0; it < iters; it += 2) {
        if constexpr (MODE == 7) { asm volatile("" ::: "memory"); B = __builtin_bit_cast(h8, s_b[it & 31][lane]); }
        asm volatile("" : "+v"(A));   // opaque: the two MFMAs of an iteration are not the same value
        D1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, zero, 0, 0, 0);
        if constexpr (MODE == 6) { tree(X); asm volatile("" : "+v"(X)); } else tree(D0);
        asm volatile("" : "+v"(A));
        if constexpr (MODE == 7) { asm volatile("" ::: "memory"); B = __builtin_bit_cast(h8, s_b[(it + 1) & 31][lane]); }
        D0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, zero, 0, 0, 0);
        if constexpr (MODE == 6) { tree(Y); asm volatile("" : "+v"(Y)); } else tree(D1);
    }
    out[blockIdx.x * 256 + threadIdx.x] = m + __float_as_int(D0[0]) + __float_as_int(D1[1]);
}

template <int MODE>
void run2(const char* what, int waves_per_simd, const uint4* d_in, int* d_out) {
    const int iters = 20000;
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    const int grid = 256 * waves_per_simd;
    probe2<MODE><<<grid, 256>>>(d_in, d_out, 100);
    (void)hipEventRecord(a);
    probe2<MODE><<<grid, 256>>>(d_in, d_out, iters);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    printf("%-44s waves/SIMD=%d : %6.2f ns per MFMA per SIMD\n", what, waves_per_simd, ms * 1e6 / (double(waves_per_simd) * iters));
}

template <int MODE, int K>
void run(const char* what, int waves_per_simd, const uint4* d_in, int* d_out) {
    const int iters = 20000;
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    const int grid = 256 * waves_per_simd;   // 256-thread blocks: one wave per SIMD each
    probe<MODE, K><<<grid, 256>>>(d_in, d_out, 100);
    (void)hipEventRecord(a);
    probe<MODE, K><<<grid, 256>>>(d_in, d_out, iters);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    const double ns = ms * 1e6 / (double(waves_per_simd) * iters);
    printf("%-44s waves/SIMD=%d : %6.2f ns per MFMA per SIMD\n", what, waves_per_simd, ns);
}

int main() {
    uint4* d_in;
    int* d_out;
    (void)hipMalloc(&d_in, 128 * sizeof(uint4));
    std::vector<uint32_t> h(128 * 4, 0x3c003c00u);
    (void)hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    (void)hipMalloc(&d_out, 256 * 8 * 256 * sizeof(int));
    for (int w : {1, 2, 4, 5}) {
        run2<5>("alternating tiles, tree on the MFMA result", w, d_in, d_out);
        run2<6>("alternating tiles, tree on other registers", w, d_in, d_out);
        run2<7>("alternating tiles, tree on result + ds_read", w, d_in, d_out);
    }
    for (int w : {2, 4}) {
        run<4, 0>("MFMA only", w, d_in, d_out);
        run<0, 8>("8 max3, one chain, on D", w, d_in, d_out);
        run<1, 8>("8 max3, tree, on D", w, d_in, d_out);
        run<2, 8>("8 max3, 4 accumulators, not on D", w, d_in, d_out);
        run<2, 4>("4 max3, 4 accumulators, not on D", w, d_in, d_out);
        run<3, 8>("8 max (2 operands), 4 accumulators", w, d_in, d_out);
        run<3, 16>("16 max (2 operands), 4 accumulators", w, d_in, d_out);
    }
    return 0;
}

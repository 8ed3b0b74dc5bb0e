// Ceilings of the LDS-bucketed permutation dot (csrc/sqgr_autocorr.hip: k_perm_dot_lds) measured on the GPU box:
//   * ds_read_b128 from a 160 KiB workgroup allocation, 16 waves per CU: conflict-free rows (lane * 16) and RANDOM
//     16-byte rows (the kernel's pattern: every lane follows its own permutation's list) -> bytes/s, clk per wave-instr
//   * ds_read_b64 random 8-byte words (the Geary row sums)
//   * v_fma_f64 issue rate
// Build + run (GPU box):  hipcc --offload-arch=gfx950 -O3 -o tools/ubench_lds_read.bin tools/ubench_lds_read.hip
//                         tools/ubench_lds_read.bin > profiles/<tag>_ubench_lds_read.json
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                                     \
    do {                                                                                             \
        hipError_t e__ = (x);                                                                        \
        if (e__ != hipSuccess) {                                                                     \
            fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e__), __FILE__, __LINE__);     \
            exit(1);                                                                                 \
        }                                                                                            \
    } while (0)

constexpr int ROWS = 10224;  // 16-byte rows in 163 584 B of LDS
constexpr int THREADS = 1024;

// MODE 0: rows lane*1 (+ register offset): conflict-free   1: pseudo-random rows   2: random 8-byte words (ds_read_b64)
// MODE 3: random rows whose class (row mod 16) is the lane's position in its ds_read_b128 service group as MI355X_MICROARCH.md lists
//   them ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32): 16 classes per group if the groups are those, 2-way collisions if the groups
//   were 16 consecutive lanes — the premise of k_bucket_order_joint.   MODE 4: random rows of class lane mod 16.
__device__ __forceinline__ uint32_t group_pos(uint32_t lane) {
    const uint32_t x = lane & 31u;
    if (x < 4) return x;
    if (x < 12) return x - 4;
    if (x < 16) return x - 8;
    if (x < 20) return x - 8;
    if (x < 28) return x - 12;
    return x - 16;
}
template <int MODE>
__global__ __launch_bounds__(THREADS) void k_lds_read(double* out, uint32_t seed, int iters) {
    extern __shared__ double2 lds2[];
    for (int i = threadIdx.x; i < ROWS; i += THREADS) lds2[i] = make_double2(1.0 + i, 0.5 * i);
    __syncthreads();
    uint32_t ad[8];
    uint32_t h = threadIdx.x * 2654435761u + seed + blockIdx.x * 40503u;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        h = h * 1664525u + 1013904223u;
        if (MODE == 0) ad[r] = ((threadIdx.x + r * THREADS) % ROWS) * 16;
        else if (MODE == 1) ad[r] = ((h >> 8) % ROWS) * 16;
        else if (MODE >= 3) ad[r] = ((((h >> 8) % (ROWS / 16)) << 4) | (MODE == 3 ? group_pos(threadIdx.x) : (threadIdx.x & 15u))) * 16;
        else ad[r] = ((h >> 8) % (ROWS * 2)) * 8;
    }
    double acc = 0.0;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 2) {
            double v[8];
            asm volatile(
                "ds_read_b64 %0, %8\n ds_read_b64 %1, %9\n ds_read_b64 %2, %10\n ds_read_b64 %3, %11\n"
                "ds_read_b64 %4, %12\n ds_read_b64 %5, %13\n ds_read_b64 %6, %14\n ds_read_b64 %7, %15\n"
                "s_waitcnt lgkmcnt(0)\n"
                : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]), "v"(ad[4]), "v"(ad[5]), "v"(ad[6]), "v"(ad[7])
                : "memory");
            acc += v[0] + v[7];
        } else {
            double2 v[8];
            asm volatile(
                "ds_read_b128 %0, %8\n ds_read_b128 %1, %9\n ds_read_b128 %2, %10\n ds_read_b128 %3, %11\n"
                "ds_read_b128 %4, %12\n ds_read_b128 %5, %13\n ds_read_b128 %6, %14\n ds_read_b128 %7, %15\n"
                "s_waitcnt lgkmcnt(0)\n"
                : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]), "v"(ad[4]), "v"(ad[5]), "v"(ad[6]), "v"(ad[7])
                : "memory");
            acc += v[0].x + v[7].y;
        }
        if (MODE != 0) {  // new pseudo-random rows every trip (cheap next to 8 reads)
            const uint32_t lim = MODE == 1 ? ROWS : ROWS * 2, sh = MODE == 1 ? 4 : 3;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                h = h * 1664525u + 1013904223u;
                ad[r] = (__umulhi(h, lim)) << sh;
                if (MODE >= 3) ad[r] = (((__umulhi(h, (uint32_t)(ROWS / 16))) << 4) | (MODE == 3 ? group_pos(threadIdx.x) : (threadIdx.x & 15u))) << 4;
            }
        }
    }
    out[blockIdx.x * THREADS + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_fma64(double* out, int iters) {
    double a[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) a[r] = 1.0 + threadIdx.x * 1e-3 + r;
    const double m = 1.0000001, c = 1e-9;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) a[r] = fma(a[r], m, c);
#pragma unroll
        for (int r = 0; r < 8; ++r) a[r] = fma(a[r], m, c);
    }
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < 8; ++r) s += a[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    CHECK(hipSetDevice(0));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double F = 2.4e9;
    double* out = nullptr;
    CHECK(hipMalloc(&out, (size_t)cus * 8 * THREADS * 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const size_t lds = (size_t)ROWS * 16;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_lds_read<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_lds_read<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_lds_read<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_lds_read<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_lds_read<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    printf("{\n  \"device\": \"%s\", \"cus\": %d, \"nominal_clock_hz\": %.3g,\n  \"lds_read\": [\n", prop.gcnArchName, cus, F);
    auto run = [&](const char* name, int mode, int bytes_per_lane, bool last) {
        const int iters = 4096, blocks = cus * 4;
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CHECK(hipEventRecord(e0, 0));
            if (mode == 0) hipLaunchKernelGGL(k_lds_read<0>, dim3(blocks), dim3(THREADS), lds, 0, out, 1u + rep, rep ? iters : 16);
            if (mode == 1) hipLaunchKernelGGL(k_lds_read<1>, dim3(blocks), dim3(THREADS), lds, 0, out, 1u + rep, rep ? iters : 16);
            if (mode == 2) hipLaunchKernelGGL(k_lds_read<2>, dim3(blocks), dim3(THREADS), lds, 0, out, 1u + rep, rep ? iters : 16);
            if (mode == 3) hipLaunchKernelGGL(k_lds_read<3>, dim3(blocks), dim3(THREADS), lds, 0, out, 1u + rep, rep ? iters : 16);
            if (mode == 4) hipLaunchKernelGGL(k_lds_read<4>, dim3(blocks), dim3(THREADS), lds, 0, out, 1u + rep, rep ? iters : 16);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipGetLastError());
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        const double winstr = (double)blocks * (THREADS / 64) * iters * 8;
        const double rate = winstr / (best * 1e-3);
        printf("    {\"pattern\": \"%s\", \"wave_instr_per_s\": %.6g, \"clk_per_wave_instr_per_cu\": %.3f, \"bytes_per_s\": %.6g}%s\n", name, rate,
               cus * F / rate, rate * 64 * bytes_per_lane, last ? "" : ",");
    };
    run("ds_read_b128 conflict-free rows (lane*16), 160 KB/workgroup, 16 waves/CU", 0, 16, false);
    run("ds_read_b128 random 16-byte rows of 160 KB (k_perm_dot_lds pattern), 16 waves/CU", 1, 16, false);
    run("ds_read_b128 random rows, row mod 16 = position in the guide's service group (16 classes per group)", 3, 16, false);
    run("ds_read_b128 random rows, row mod 16 = lane mod 16", 4, 16, false);
    run("ds_read_b64 random 8-byte words of 160 KB, 16 waves/CU", 2, 8, true);
    printf("  ],\n");
    {
        const int iters = 4096, blocks = cus * 8;
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_fma64, dim3(blocks), dim3(256), 0, 0, out, rep ? iters : 16);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        const double winstr = (double)blocks * 4 * iters * 16;
        const double rate = winstr / (best * 1e-3);
        printf("  \"valu\": [{\"op\": \"v_fma_f64\", \"wave_instr_per_s\": %.6g, \"clk_per_wave_instr_per_simd\": %.3f}]\n}\n", rate, cus * 4 * F / rate);
    }
    return 0;
}

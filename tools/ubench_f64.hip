// Issue rates of the float64 VALU instructions behind Ripley's pair / nearest-neighbour kernels and the Moran / Geary dots
// (task 1a: "float64 VALU rates are not in the µbench yet").  Same method as tools/ubench_ops.hip: 8 independent
// dependency chains of ONE instruction per lane, 32 instructions per loop trip, 8 waves per SIMD on every CU; reported as
// wave-instructions/s for the chip, clk per wave-instruction per SIMD at the nominal 2.4 GHz, and cost relative to v_fma_f32
// measured in the same process.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_f64.bin tools/ubench_f64.hip && tools/ubench_f64.bin profiles/r03_ubench_f64.json
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(x)                                                                                 \
    do {                                                                                         \
        hipError_t e__ = (x);                                                                    \
        if (e__ != hipSuccess) {                                                                 \
            fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e__), __FILE__, __LINE__); \
            exit(1);                                                                             \
        }                                                                                        \
    } while (0)

#define REP8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define F64_KERNEL(NAME, OP)                                                                                                 \
    __global__ __launch_bounds__(256) void NAME(double* out, double seed, int iters) {                                       \
        double a0 = threadIdx.x * 1.0009765625 + seed, a1 = a0 * 0.5 + 1, a2 = a0 * 0.25 + 7, a3 = a0 + 3, a4 = a0 + 77,      \
               a5 = a1 * 0.75, a6 = a2 + a1, a7 = a3 + a2;                                                                    \
        double b = 0.999999, c = 1e-9 + seed * 1e-12;                                                                        \
        uint32_t m0 = 0, m1 = 0, m2 = 0, m3 = 0, m4 = 0, m5 = 0, m6 = 0, m7 = 0;                                             \
        for (int i = 0; i < iters; ++i) {                                                                                    \
            asm volatile(REP8(OP) REP8(OP) REP8(OP) REP8(OP)                                                                 \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(m0), "+v"(m1), \
                           "+v"(m2), "+v"(m3), "+v"(m4), "+v"(m5), "+v"(m6), "+v"(m7)                                        \
                         : "v"(b), "v"(c)                                                                                    \
                         : "vcc");                                                                                           \
        }                                                                                                                    \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (double)(m0 ^ m1 ^ m2 ^ m3 ^ m4 ^ m5 ^ m6 ^ m7); \
    }
// operands: %0..%7 doubles (accumulators), %8..%15 32-bit side registers, %16 = b, %17 = c
#define OP_FMA64(d) "v_fma_f64 %" #d ", %" #d ", %16, %17\n"
#define OP_MUL64(d) "v_mul_f64 %" #d ", %" #d ", %16\n"
#define OP_ADD64(d) "v_add_f64 %" #d ", %" #d ", %17\n"
#define OP_MIN64(d) "v_min_f64 %" #d ", %" #d ", %16\n"
#define OP_MAX64(d) "v_max_f64 %" #d ", %" #d ", %17\n"
#define OP_CMP64_0(d) "v_cmp_le_f64 vcc, %" #d ", %16\n v_addc_co_u32 %8, vcc, %8, 0, vcc\n"
#define OP_CMP64(d) "v_cmp_le_f64 vcc, %" #d ", %16\n"
#define OP_CVT64(d) "v_cvt_i32_f64 %8, %" #d "\n"
#define OP_FMA32(d) "v_fma_f32 %8, %8, %9, %10\n"
#define OP_SQRT64(d) "v_sqrt_f64 %" #d ", %" #d "\n"
#define OP_RCP64(d) "v_rcp_f64 %" #d ", %" #d "\n"

F64_KERNEL(k_fma64, OP_FMA64)
F64_KERNEL(k_mul64, OP_MUL64)
F64_KERNEL(k_add64, OP_ADD64)
F64_KERNEL(k_min64, OP_MIN64)
F64_KERNEL(k_max64, OP_MAX64)
F64_KERNEL(k_cmp64, OP_CMP64)
F64_KERNEL(k_cmpaddc64, OP_CMP64_0)
F64_KERNEL(k_cvt64, OP_CVT64)
F64_KERNEL(k_sqrt64, OP_SQRT64)
F64_KERNEL(k_rcp64, OP_RCP64)

// the f32 reference in the same harness (8 chains on the 32-bit side registers)
__global__ __launch_bounds__(256) void k_fma32(double* out, double seed, int iters) {
    float a0 = threadIdx.x + (float)seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 7, a3 = a0 + 9, a4 = a0 + 77, a5 = a1 * 9, a6 = a2 + a1, a7 = a3 + a2;
    const float b = 0.9999f, c = 1e-3f;
    for (int i = 0; i < iters; ++i) {
#define F(d) "v_fma_f32 %" #d ", %" #d ", %8, %9\n"
        asm volatile(REP8(F) REP8(F) REP8(F) REP8(F) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
#undef F
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

int main(int argc, char** argv) {
    CHECK(hipSetDevice(0));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, blocks = cus * 8;
    double* out = nullptr;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    struct Row { std::string name; double rate, clk; };
    std::vector<Row> rows;
    auto run = [&](const char* name, void (*kern)(double*, double, int), int per_trip) {
        const int iters = 2048;
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1.0, 32);
        CHECK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1.0 + rep, iters);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipGetLastError());
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double winstr = (double)blocks * 4 * iters * per_trip;
        const double rate = winstr / (best * 1e-3);
        rows.push_back({name, rate, cus * 4 * 2.4e9 / rate});
    };
    run("v_fma_f32", k_fma32, 32);
    run("v_fma_f64", k_fma64, 32);
    run("v_mul_f64", k_mul64, 32);
    run("v_add_f64", k_add64, 32);
    run("v_min_f64", k_min64, 32);
    run("v_max_f64", k_max64, 32);
    run("v_cmp_le_f64", k_cmp64, 32);
    run("v_cmp_le_f64+v_addc_co_u32 (pair)", k_cmpaddc64, 64);
    run("v_cvt_i32_f64", k_cvt64, 32);
    run("v_sqrt_f64", k_sqrt64, 32);
    run("v_rcp_f64", k_rcp64, 32);
    std::string js = "{\n  \"device\": \"" + std::string(prop.gcnArchName) + "\", \"cus\": " + std::to_string(cus) + ", \"nominal_clock_hz\": 2.4e9,\n  \"valu\": [\n";
    for (size_t i = 0; i < rows.size(); ++i) {
        char buf[512];
        snprintf(buf, sizeof(buf), "    {\"op\": \"%s\", \"wave_instr_per_s\": %.6g, \"clk_per_wave_instr_per_simd\": %.3f, \"cost_vs_v_fma_f32\": %.3f}%s\n",
                 rows[i].name.c_str(), rows[i].rate, rows[i].clk, rows[0].rate / rows[i].rate, i + 1 < rows.size() ? "," : "");
        js += buf;
        printf("%-40s %14.4g wave-instr/s %8.2f clk/SIMD\n", rows[i].name.c_str(), rows[i].rate, rows[i].clk);
    }
    js += "  ]\n}\n";
    if (argc > 1) {
        FILE* f = fopen(argv[1], "w");
        if (f) {
            fputs(js.c_str(), f);
            fclose(f);
        }
    }
    return 0;
}

// LDS-instruction mix of the pair-histogram kernels (co_occurrence: ds_read_u16 table look-up + ds_read2_b32 of two adjacent
// thresholds + ds_add_u32 into a per-lane histogram column per pair; Ripley L: the same with ds_read2_b64): what does one such
// triple cost per CU when nothing else runs?  bench.py prices k_cooccur_fast's DS instruction count (PMC) against this rate next to
// its VALU-mix fraction — whichever is higher names the bound.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_ds_mix.bin tools/ubench_ds_mix.hip && tools/ubench_ds_mix.bin profiles/r03_ubench_ds_mix.json
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>

#define CHECK(x)                                                                                 \
    do {                                                                                         \
        hipError_t e__ = (x);                                                                    \
        if (e__ != hipSuccess) {                                                                 \
            fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e__), __FILE__, __LINE__); \
            exit(1);                                                                             \
        }                                                                                        \
    } while (0)

constexpr int L = 49, TRASH = 3, HC = 64, NCELLS = 4096;

// MODE bit 0: table look-up (ds_read_u16), bit 1: threshold pair (ds_read2_b32 / WIDE: ds_read2_b64), bit 2: ds_add_u32
// STAG: the thresholds are stored 32 times, copy c in bank (pair) c, and lane l reads copy l & 31 (what the kernels do since round 3);
// else the plain [L + 2] array (round 2: random bins of 64 lanes meet in the banks)
template <int MODE, bool WIDE, bool STAG = true>
__global__ __launch_bounds__(256) void k_ds(uint32_t* out, int iters) {
    extern __shared__ uint32_t smem[];
    uint32_t* hist = smem;                                                   // [L + TRASH][HC]
    double* thr = reinterpret_cast<double*>(smem + (L + TRASH) * HC);        // [L + 2][32] doubles (WIDE) or floats
    uint16_t* cell = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(thr) + (size_t)(L + 2) * 32 * (WIDE ? 8 : 4));  // [NCELLS]
    const int t = threadIdx.x;
    const int TS = STAG ? 32 : 1, tl = STAG ? (t & 31) : 0;
    for (int i = t; i < (L + TRASH) * HC; i += 256) hist[i] = 0;
    for (int i = t; i < (L + 2) * 32; i += 256) {
        if (WIDE) thr[i] = (double)i;
        else reinterpret_cast<float*>(thr)[i] = (float)i;
    }
    for (int i = t; i < NCELLS; i += 256) cell[i] = (uint16_t)((i * (L - 2)) / NCELLS);
    __syncthreads();
    uint32_t h = (blockIdx.x * 256u + t) * 2654435761u + 7u, acc = 0;
    uint32_t* my = hist + (t & (HC - 1));
    const float* thr32 = reinterpret_cast<const float*>(thr);
    for (int it = 0; it < iters; ++it) {
        int g[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            h = h * 1664525u + 1013904223u;
            const uint32_t c = h >> 20;  // random cell of 4096
            g[u] = (MODE & 1) ? (int)cell[c] : (int)(c & 31u);
        }
        float a0[8], a1[8];
        double b0[8], b1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE & 2) {
                if (WIDE) { b0[u] = thr[g[u] * TS + tl]; b1[u] = thr[(g[u] + 1) * TS + tl]; }
                else { a0[u] = thr32[g[u] * TS + tl]; a1[u] = thr32[(g[u] + 1) * TS + tl]; }
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            int gg = g[u];
            if (MODE & 2) gg += WIDE ? (int)(b0[u] > 1e9) + (int)(b1[u] > 1e9) : (int)(a0[u] > 1e9f) + (int)(a1[u] > 1e9f);
            if (MODE & 4) atomicAdd(my + gg * HC, 1u);
            else acc += (uint32_t)gg;
        }
    }
    __syncthreads();
    out[blockIdx.x * 256 + t] = acc + hist[t];
}

int main(int argc, char** argv) {
    CHECK(hipSetDevice(0));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint32_t* out = nullptr;
    CHECK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const size_t lds_narrow = (size_t)(L + TRASH) * HC * 4 + (size_t)(L + 2) * 32 * 4 + NCELLS * 2;
    const size_t lds_wide = (size_t)(L + TRASH) * HC * 4 + (size_t)(L + 2) * 32 * 8 + NCELLS * 2;
    std::string js = "{\n  \"device\": \"" + std::string(prop.gcnArchName) + "\", \"cus\": " + std::to_string(cus) +
                     ",\n  \"note\": \"clk at the nominal 2.4 GHz per PAIR STEP of one wave (the DS instructions named) per CU\",\n  \"ds_mix\": [\n";
    bool first = true;
    auto run = [&](const char* name, void (*kern)(uint32_t*, int), int blocks_per_cu, int n_ds, bool wide = false) {
        const size_t lds = wide ? lds_wide : lds_narrow;
        if ((size_t)blocks_per_cu * lds > 160 * 1024) return;
        const int iters = 4096, blocks = cus * blocks_per_cu;
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, out, 16);
        CHECK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, out, iters);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipGetLastError());
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double steps = (double)blocks * 4 * iters * 8;  // wave pair-steps
        const double clk = cus * 2.4e9 * (best * 1e-3) / steps;
        char buf[512];
        snprintf(buf, sizeof(buf), "%s    {\"mix\": \"%s\", \"waves_per_cu\": %d, \"ds_instr_per_step\": %d, \"clk_per_step_per_cu\": %.2f, \"ds_wave_instr_per_s\": %.6g}",
                 first ? "" : ",\n", name, blocks_per_cu * 4, n_ds, clk, steps * n_ds / (best * 1e-3));
        js += buf;
        printf("%-70s %2d waves/CU  %.2f clk per step per CU\n", name, blocks_per_cu * 4, clk);
        first = false;
    };
    for (int bpc : {2, 4, 5}) {
        run("ds_add_u32 (per-lane column)", k_ds<4, false>, bpc, 1);
        run("ds_read_u16 (random table cell)", k_ds<1, false>, bpc, 1);
        run("co_occurrence: ds_read_u16 + ds_read2_b32 + ds_add_u32, plain threshold array (round 2)", k_ds<7, false, false>, bpc, 3);
        run("co_occurrence: ds_read_u16 + ds_read2_b32 + ds_add_u32", k_ds<7, false, true>, bpc, 3);
        run("ripley L: ds_read_u16 + ds_read2_b64 + ds_add_u32, plain threshold array (round 2)", k_ds<7, true, false>, bpc, 3, true);
        run("ripley L: ds_read_u16 + ds_read2_b64 + ds_add_u32", k_ds<7, true, true>, bpc, 3, true);
    }
    js += "\n  ]\n}\n";
    if (argc > 1) {
        FILE* f = fopen(argv[1], "w");
        if (f) {
            fputs(js.c_str(), f);
            fclose(f);
        }
    }
    return 0;
}

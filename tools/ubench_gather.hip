// How fast does a CU gather 16-byte rows out of a 16 MB table (L2 / Infinity-Cache resident), and does it matter how many
// lanes share a row?  The nhood count kernel reads one 16-byte label row per (edge endpoint, batch of 16 permutations):
//   MODE 0: a QUAD of lanes reads one row, 4 bytes per lane (global_load_dword)      — what k_count does
//   MODE 1: a PAIR of lanes reads one row, 8 bytes per lane (global_load_dwordx2)
//   MODE 2: ONE lane reads one row, 16 bytes (global_load_dwordx4)
// each with random rows (LOCAL = 0) and with the rows of neighbouring requesters adjacent (LOCAL = 1: request r of a wave
// reads row base + r, the hex-grid edge lists look like that).  Output: rows/s and clk per wave-instruction per CU.
// Build + run (GPU box):  hipcc --offload-arch=gfx950 -O3 -o tools/ubench_gather.bin tools/ubench_gather.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                                 \
    do {                                                                                         \
        hipError_t e__ = (x);                                                                    \
        if (e__ != hipSuccess) {                                                                 \
            fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e__), __FILE__, __LINE__); \
            exit(1);                                                                             \
        }                                                                                        \
    } while (0)

constexpr uint32_t NROWS = 1u << 20;  // 16 MB of 16-byte rows

template <int MODE, int LOCAL>
__global__ __launch_bounds__(1024) void k_gather(const uint8_t* __restrict__ tab, uint32_t* __restrict__ out, uint32_t seed, int iters) {
    constexpr int LANES = MODE == 0 ? 4 : (MODE == 1 ? 2 : 1);  // lanes per row
    const uint32_t tid = threadIdx.x, req = tid / LANES, sub = tid % LANES;
    uint32_t h = (blockIdx.x * 1024u + (LOCAL ? (tid >> 6) : req)) * 2654435761u + seed;
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        uint32_t rows[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            h = h * 1664525u + 1013904223u;
            const uint32_t base = (h >> 7) & (NROWS - 1);
            rows[u] = LOCAL ? ((base + (req & (64 / LANES - 1))) & (NROWS - 1)) : base;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint8_t* p = tab + (size_t)rows[u] * 16 + sub * (16 / LANES);
            if (MODE == 0) acc += *reinterpret_cast<const uint32_t*>(p);
            if (MODE == 1) {
                const uint2 v = *reinterpret_cast<const uint2*>(p);
                acc += v.x ^ v.y;
            }
            if (MODE == 2) {
                const uint4 v = *reinterpret_cast<const uint4*>(p);
                acc += v.x ^ v.y ^ v.z ^ v.w;
            }
        }
    }
    out[blockIdx.x * 1024 + tid] = acc;
}

// L1-resident table (16 KB): what drives the cost of a gather instruction in the address/L1 pipeline?
//   PAT 0: all 16 quads of a wave read the SAME row        1: 16 adjacent rows (2 lines of 128 B)
//   PAT 2: 16 rows 64 B apart (16 half lines)              3: 16 rows 128 B apart (16 lines)
//   PAT 4: lane x dwordx4, 64 adjacent rows (8 lines)      5: lane x dwordx4, 64 rows 128 B apart
template <int PAT>
__global__ __launch_bounds__(1024) void k_l1(const uint8_t* __restrict__ tab, uint32_t* __restrict__ out, int iters) {
    const uint32_t tid = threadIdx.x, lane = tid & 63, quad = lane >> 2, sub = lane & 3;
    uint32_t off[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const uint32_t w = ((tid >> 6) * 8 + u) * 16;  // rows: every wave / load its own window of the table
        uint32_t row;
        if (PAT == 0) row = w;
        else if (PAT == 1) row = w + quad;
        else if (PAT == 2) row = w + quad * 4;
        else if (PAT == 3) row = w + quad * 8;
        else if (PAT == 4) row = w + lane;
        else row = w + lane * 8;
        off[u] = ((row * 16) & 16383u) + (PAT >= 4 ? 0 : sub * 4);
    }
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (PAT >= 4) {
                const uint4 v = *reinterpret_cast<const uint4*>(tab + off[u]);
                acc += v.x ^ v.y ^ v.z ^ v.w;
            } else {
                acc += *reinterpret_cast<const uint32_t*>(tab + off[u]);
            }
            off[u] = (off[u] + 2048) & 16383u;
        }
    }
    out[blockIdx.x * 1024 + tid] = acc;
}

int main() {
    CHECK(hipSetDevice(0));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, blocks = cus * 2;
    uint8_t* tab = nullptr;
    uint32_t* out = nullptr;
    CHECK(hipMalloc(&tab, (size_t)NROWS * 16));
    CHECK(hipMemset(tab, 1, (size_t)NROWS * 16));
    CHECK(hipMalloc(&out, (size_t)blocks * 1024 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("{\n  \"device\": \"%s\", \"cus\": %d, \"table_bytes\": %zu,\n  \"gather\": [\n", prop.gcnArchName, cus, (size_t)NROWS * 16);
    auto run = [&](const char* name, void (*kern)(const uint8_t*, uint32_t*, uint32_t, int), int lanes, bool last) {
        const int iters = 256;
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), 0, 0, tab, out, 1u + rep, rep ? iters : 8);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipGetLastError());
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        const double winstr = (double)blocks * 16 * iters * 8;
        const double rows = winstr * 64 / lanes;
        printf("    {\"pattern\": \"%s\", \"rows_per_s\": %.6g, \"clk_per_wave_instr_per_cu\": %.2f, \"rows_per_clk_per_cu\": %.3f}%s\n", name,
               rows / (best * 1e-3), cus * 2.4e9 / (winstr / (best * 1e-3)), rows / (best * 1e-3) / (cus * 2.4e9), last ? "" : ",");
    };
    run("quad x dword, random rows", k_gather<0, 0>, 4, false);
    run("pair x dwordx2, random rows", k_gather<1, 0>, 2, false);
    run("lane x dwordx4, random rows", k_gather<2, 0>, 1, false);
    run("quad x dword, adjacent rows per wave", k_gather<0, 1>, 4, false);
    run("pair x dwordx2, adjacent rows per wave", k_gather<1, 1>, 2, false);
    run("lane x dwordx4, adjacent rows per wave", k_gather<2, 1>, 1, true);
    printf("  ],\n  \"l1_resident\": [\n");
    auto run1 = [&](const char* name, void (*kern)(const uint8_t*, uint32_t*, int), bool last) {
        const int iters = 2048;
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), 0, 0, tab, out, rep ? iters : 8);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipGetLastError());
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        const double winstr = (double)blocks * 16 * iters * 8;
        printf("    {\"pattern\": \"%s\", \"wave_instr_per_s\": %.6g, \"clk_per_wave_instr_per_cu\": %.2f}%s\n", name, winstr / (best * 1e-3),
               cus * 2.4e9 / (winstr / (best * 1e-3)), last ? "" : ",");
    };
    run1("quad x dword, 16 quads one row", k_l1<0>, false);
    run1("quad x dword, 16 adjacent rows (2 lines)", k_l1<1>, false);
    run1("quad x dword, 16 rows 64 B apart", k_l1<2>, false);
    run1("quad x dword, 16 rows 128 B apart (16 lines)", k_l1<3>, false);
    run1("lane x dwordx4, 64 adjacent rows (8 lines)", k_l1<4>, false);
    run1("lane x dwordx4, 64 rows 128 B apart (64 lines)", k_l1<5>, true);
    printf("  ]\n}\n");
    return 0;
}

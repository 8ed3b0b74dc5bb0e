// What does the vector-memory path charge the nhood count kernel for?  (task 3.)
//
// k_count issues, per lane and stage of 4 edges: 8 label-row gathers (global_load_dword, 4 lanes share a 16-byte row,
// a wave instruction touches 16 rows) and 16 ds_add_u32.  Its probe variants showed 0.355 ms (no gathers), 0.48 ms (no
// atomics), 0.47 ms (both) per 1024 permutations.  This micro-benchmark rebuilds that shape with one factor varied at a time:
//
//   mix      G gathers + A ds_add_u32 per loop trip, the table L1-resident: is the cost of the two instruction kinds
//            additive (a shared issue/operand path) or the maximum of the two (independent pipes)?
//   spread   the 16 rows of one gather instruction packed into 2 / 4 / 8 / 16 lines of 128 B (L1-resident)
//   stream   the same gathers walking a 256 MiB table once (every row exactly once, like the slab of one launch): what the
//            compulsory misses cost at 8 / 16 / 24 loads in flight per wave
//
// Build + run (GPU box):  hipcc --offload-arch=gfx950 -O3 -o tools/ubench_count_shape.bin tools/ubench_count_shape.hip
//                         tools/ubench_count_shape.bin > gpurun_out/ubench_count_shape.json
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

constexpr int HIST_WORDS = 900 * 16;  // K = 30: the count kernel's 57.6 KB histogram -> two blocks per CU

// G gathers (rows of a window of `win_rows` rows starting at this block's slice; the window advances by `adv` rows per trip)
// and A ds_add_u32 per loop trip.  SPREAD: rows of one instruction = base + (4 * quad + u) * SPREAD / 3 (SPREAD = 1: the
// ~21 adjacent rows of 64 consecutive half-edges of a hex grid; 3: 64 rows = 8 lines; 6: 16 lines).
// ADDR: 0 64-bit per-lane address, dword per lane, 4 lanes per row   1 scalar base + 32-bit offset (one address dword), same shape
//       2 scalar base + 32-bit offset, dwordx2 per lane, 2 lanes per row (32 rows per instruction)
//       3 scalar base + 32-bit offset, dwordx4 per lane, 1 lane per row (64 rows per instruction)
template <int G, int A, int SPREAD, int ADDR = 0>
__global__ __launch_bounds__(1024, 8) void k_mix(const uint8_t* __restrict__ tab, uint32_t* __restrict__ out, uint32_t win_rows,
                                                 uint32_t adv, uint32_t rows_total, int iters) {
    extern __shared__ uint32_t hist[];
    const uint32_t tid = threadIdx.x, lane = tid & 63, quad = lane >> 2, sub = lane & 3, wave = tid >> 6;
    for (int i = tid; i < HIST_WORDS; i += 1024) hist[i] = 0;
    __syncthreads();
    // this block's slice of the table
    const uint32_t slice = (uint32_t)(((uint64_t)blockIdx.x * rows_total) / gridDim.x);
    uint32_t base = wave * (win_rows / 16);  // every wave its own part of the window
    uint32_t h = (blockIdx.x * 1024u + tid) * 2654435761u + 12345u;
    uint32_t acc = 0;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)hist;
    const uint32_t bank = lds_base + ((sub * 4 + (quad & 3)) * 4);
    for (int it = 0; it < iters; ++it) {
        uint32_t v[G > 0 ? G : 1];
#pragma unroll
        for (int u = 0; u < G; ++u) {
            if constexpr (ADDR == 0) {
                const uint32_t r = (base + ((4 * quad + (u & 3)) * SPREAD) / 3 + (u >> 2) * (win_rows / 2)) & (win_rows - 1);  // win_rows = 2^k
                v[u] = *reinterpret_cast<const uint32_t*>(tab + ((size_t)(slice + r) * 16 + sub * 4));
            } else if constexpr (ADDR == 1) {
                const uint32_t r = (base + ((4 * quad + (u & 3)) * SPREAD) / 3 + (u >> 2) * (win_rows / 2)) & (win_rows - 1);
                v[u] = *reinterpret_cast<const uint32_t*>(tab + (size_t)slice * 16 + (r * 16u + sub * 4u));
            } else if constexpr (ADDR == 2) {
                const uint32_t r = (base + ((2 * (lane >> 1) + (u & 1)) * SPREAD) / 3 + (u >> 1) * (win_rows / 2)) & (win_rows - 1);
                const uint2 t = *reinterpret_cast<const uint2*>(tab + (size_t)slice * 16 + (r * 16u + (lane & 1u) * 8u));
                v[u] = t.x ^ t.y;
            } else {
                const uint32_t r = (base + (lane * SPREAD) / 3 + u * (win_rows / 2)) & (win_rows - 1);
                const uint4 t = *reinterpret_cast<const uint4*>(tab + (size_t)slice * 16 + r * 16u);
                v[u] = t.x ^ t.y ^ t.z ^ t.w;
            }
        }
#pragma unroll
        for (int s = 0; s < A; ++s) {
            h = h * 1664525u + 1013904223u;
            const uint32_t addr = ((((h >> 16) * 900u) >> 10) & ~63u) + bank;  // pair = floor(u16 * 900 / 2^16), 64 bytes of counters per pair
            asm volatile("ds_add_u32 %0, %1" : : "v"(addr), "v"(1u) : "memory");
        }
#pragma unroll
        for (int u = 0; u < G; ++u) acc += v[u];
        base = (base + adv) & (win_rows - 1);
    }
    __syncthreads();
    out[blockIdx.x * 1024 + tid] = acc + hist[tid];
}

// streaming: each wave walks its own contiguous run of rows exactly once, D gather instructions in flight per wave
template <int D>
__global__ __launch_bounds__(1024, 8) void k_stream(const uint8_t* __restrict__ tab, uint32_t* __restrict__ out, uint64_t rows_total) {
    const uint32_t tid = threadIdx.x, lane = tid & 63, quad = lane >> 2, sub = lane & 3;
    const uint64_t nwaves = (uint64_t)gridDim.x * 16, w = (uint64_t)blockIdx.x * 16 + (tid >> 6);
    const uint64_t r0 = w * (rows_total / nwaves), r1 = r0 + rows_total / nwaves;
    uint32_t acc = 0;
    for (uint64_t r = r0; r + 16 * D <= r1; r += 16 * D) {
        uint32_t v[D];
#pragma unroll
        for (int u = 0; u < D; ++u) v[u] = *reinterpret_cast<const uint32_t*>(tab + ((r + u * 16 + quad) * 16 + sub * 4));
#pragma unroll
        for (int u = 0; u < D; ++u) acc += v[u];
    }
    out[blockIdx.x * 1024 + tid] = acc;
}

int main() {
    CHECK(hipSetDevice(0));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, blocks = cus * 2;
    const uint64_t table_rows = (uint64_t)1 << 24;  // 256 MiB of 16-byte rows
    uint8_t* tab = nullptr;
    uint32_t* out = nullptr;
    CHECK(hipMalloc(&tab, table_rows * 16));
    CHECK(hipMemset(tab, 1, table_rows * 16));
    CHECK(hipMalloc(&out, (size_t)blocks * 1024 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const size_t lds = HIST_WORDS * 4;
    printf("{\n  \"device\": \"%s\", \"cus\": %d,\n  \"note\": \"clk at the nominal 2.4 GHz per loop trip of ONE wave, per CU (32 waves per CU resident)\",\n  \"mix\": [\n", prop.gcnArchName, cus);
    bool first = true;
    auto time_it = [&](auto launch) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CHECK(hipEventRecord(e0, 0));
            launch(rep == 0);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipGetLastError());
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        return best;
    };
    auto mix = [&](const char* name, auto kern, int G, int A, uint32_t win_rows, uint32_t adv, uint32_t rows_total) {
        const int iters = 4096;
        const float ms = time_it([&](bool warm) { hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), lds, 0, tab, out, win_rows, adv, rows_total, warm ? 16 : iters); });
        const double trips = (double)blocks * 16 * iters;  // wave trips
        const double clk_per_trip_cu = cus * 2.4e9 * (ms * 1e-3) / trips;
        printf("%s    {\"case\": \"%s\", \"gathers\": %d, \"ds_add\": %d, \"clk_per_wave_trip_per_cu\": %.2f, \"clk_per_gather\": %.2f, \"clk_per_ds_add\": %.2f}",
               first ? "" : ",\n", name, G, A, clk_per_trip_cu, G ? clk_per_trip_cu / G : 0.0, A ? clk_per_trip_cu / A : 0.0);
        first = false;
    };
    // L1-resident: a 512-row (8 KB) window per block, not advancing
    mix("L1 hit, adjacent rows: 8 gathers", k_mix<8, 0, 1>, 8, 0, 512, 0, 512 * (uint32_t)blocks);
    mix("16 ds_add only", k_mix<0, 16, 1>, 0, 16, 512, 0, 512 * (uint32_t)blocks);
    mix("L1 hit, adjacent rows: 8 gathers + 16 ds_add", k_mix<8, 16, 1>, 8, 16, 512, 0, 512 * (uint32_t)blocks);
    mix("L1 hit, adjacent rows: 4 gathers + 16 ds_add", k_mix<4, 16, 1>, 4, 16, 512, 0, 512 * (uint32_t)blocks);
    mix("L1 hit, adjacent rows: 16 gathers + 16 ds_add", k_mix<16, 16, 1>, 16, 16, 512, 0, 512 * (uint32_t)blocks);
    mix("L1 hit, adjacent rows: 8 gathers + 8 ds_add", k_mix<8, 8, 1>, 8, 8, 512, 0, 512 * (uint32_t)blocks);
    mix("L1 hit, 8 lines per gather: 8 gathers", k_mix<8, 0, 3>, 8, 0, 512, 0, 512 * (uint32_t)blocks);
    mix("L1 hit, 8 lines per gather: 8 gathers + 16 ds_add", k_mix<8, 16, 3>, 8, 16, 512, 0, 512 * (uint32_t)blocks);
    mix("L1 hit, 16 lines per gather: 8 gathers", k_mix<8, 0, 6>, 8, 0, 512, 0, 512 * (uint32_t)blocks);
    mix("L1 hit, 16 lines per gather: 8 gathers + 16 ds_add", k_mix<8, 16, 6>, 8, 16, 512, 0, 512 * (uint32_t)blocks);
    // the count kernel's reuse: every row is touched by ~12 gather instructions before the window has moved past it
    // (3 half-edges as `a`, 3 as `b`, 4 instructions of a stage touch the same rows) -> window advances 21 rows per 8 gathers x 4
    mix("sliding window over 256 MiB (count-kernel reuse): 8 gathers", k_mix<8, 0, 1>, 8, 0, (uint32_t)(table_rows / blocks), 6, (uint32_t)table_rows);
    mix("sliding window over 256 MiB (count-kernel reuse): 8 gathers + 16 ds_add", k_mix<8, 16, 1>, 8, 16, (uint32_t)(table_rows / blocks), 6, (uint32_t)table_rows);
    // the same 64 edges' rows per wave trip fetched three ways (L1-resident), each beside the 16 ds_add of a stage
    mix("address form: 8 x dword, 64-bit address + 16 ds_add", k_mix<8, 16, 1, 0>, 8, 16, 512, 0, 512 * (uint32_t)blocks);
    mix("address form: 8 x dword, scalar base + 32-bit offset + 16 ds_add", k_mix<8, 16, 1, 1>, 8, 16, 512, 0, 512 * (uint32_t)blocks);
    mix("address form: 4 x dwordx2 (2 lanes per row), scalar base + 16 ds_add", k_mix<4, 16, 1, 2>, 4, 16, 512, 0, 512 * (uint32_t)blocks);
    mix("address form: 2 x dwordx4 (lane per row), scalar base + 16 ds_add", k_mix<2, 16, 1, 3>, 2, 16, 512, 0, 512 * (uint32_t)blocks);
    mix("address form: 8 x dword, scalar base, alone", k_mix<8, 0, 1, 1>, 8, 0, 512, 0, 512 * (uint32_t)blocks);
    mix("address form: 4 x dwordx2, scalar base, alone", k_mix<4, 0, 1, 2>, 4, 0, 512, 0, 512 * (uint32_t)blocks);
    mix("address form: 2 x dwordx4, scalar base, alone", k_mix<2, 0, 1, 3>, 2, 0, 512, 0, 512 * (uint32_t)blocks);
    printf("\n  ],\n  \"stream\": [\n");
    first = true;
    auto stream = [&](const char* name, auto kern, int D) {
        const float ms = time_it([&](bool) { hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), 0, 0, tab, out, table_rows); });
        const double bytes = (double)table_rows * 16;
        const double winstr = bytes / 256.0;
        printf("%s    {\"case\": \"%s\", \"in_flight_per_wave\": %d, \"GBps\": %.1f, \"clk_per_wave_instr_per_cu\": %.2f}", first ? "" : ",\n", name, D,
               bytes / (ms * 1e-3) / 1e9, cus * 2.4e9 * (ms * 1e-3) / winstr);
        first = false;
    };
    stream("quad x dword rows read once, 256 MiB", k_stream<8>, 8);
    stream("quad x dword rows read once, 256 MiB", k_stream<16>, 16);
    stream("quad x dword rows read once, 256 MiB", k_stream<24>, 24);
    printf("\n  ]\n}\n");
    return 0;
}

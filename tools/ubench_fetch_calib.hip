// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of the nhood kernels.
// MI355X_MICROARCH.md §HBM: FETCH_SIZE reports exactly half of the bytes of a wide (16 B per lane) coalesced streaming read;
// "other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern".
// Every kernel below moves a KNOWN number of bytes (512 MiB, twice the Infinity Cache) exactly once:
//   k_calib_read_b16      16 B per lane, coalesced                        (the guide's calibrated case: expect 0.5)
//   k_calib_read_quadrow  4 B per lane, 4 lanes per 16-byte row, 16 rows per wave instruction  (k_count's label-row gathers)
//   k_calib_read_pair8    8 B per lane, coalesced                         (k_count's edge-list loads)
//   k_calib_write_row16   one 16-byte row per lane                        (k_shuffle's slab stores)
//   k_calib_write_b4      4 B per lane, coalesced                         (k_count's partial histograms)
// Round 4: does any counter tell Infinity-Cache (MALL) hits from DRAM reads?
//   k_calib_reread_24m    one 24 MiB buffer (the size of nhood's half edge list) read 160 times by the whole chip: 3.75 GiB
//                         cross the fabric, 24 MiB are compulsory DRAM reads if the 256 MiB Infinity Cache holds the buffer
//   k_calib_stream_2g5    one 2.5 GiB buffer (the size of a 160-batch label slab) read once: fabric bytes = DRAM bytes
// Both also timed with HIP events: a re-read rate above what HBM can deliver is on-die traffic whatever the counters say.
// Run under `rocprofv3 --pmc FETCH_SIZE`, `--pmc WRITE_SIZE` and `--pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum
// TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum` (tools/profile_round.sh does, and divides).
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

constexpr uint64_t BYTES = (uint64_t)512 << 20;

__global__ __launch_bounds__(256) void k_calib_read_b16(const uint4* __restrict__ src, uint32_t* __restrict__ out, uint64_t n16) {
    uint32_t acc = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256) {
        const uint4 v = src[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// wave w reads rows [w * per, (w + 1) * per): instruction u of a trip reads the 16 rows r + 16 u + quad, 4 B per lane
__global__ __launch_bounds__(256) void k_calib_read_quadrow(const uint8_t* __restrict__ src, uint32_t* __restrict__ out, uint64_t rows) {
    const uint32_t lane = threadIdx.x & 63, quad = lane >> 2, sub = lane & 3;
    const uint64_t nw = (uint64_t)gridDim.x * 4, w = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), per = rows / nw;
    uint32_t acc = 0;
    for (uint64_t r = w * per; r < (w + 1) * per; r += 64) {
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += *reinterpret_cast<const uint32_t*>(src + ((r + 16 * u + quad) * 16 + sub * 4));
    }
    if (acc == 0x12345678u) out[0] = acc;
}

__global__ __launch_bounds__(256) void k_calib_read_pair8(const uint2* __restrict__ src, uint32_t* __restrict__ out, uint64_t n8) {
    uint32_t acc = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)256 + threadIdx.x; i < n8; i += (uint64_t)gridDim.x * 256) {
        const uint2 v = src[i];
        acc += v.x ^ v.y;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

__global__ __launch_bounds__(256) void k_calib_write_row16(uint4* __restrict__ dst, uint64_t n16) {
    for (uint64_t i = blockIdx.x * (uint64_t)256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256)
        dst[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}

__global__ __launch_bounds__(256) void k_calib_write_b4(uint32_t* __restrict__ dst, uint64_t n4) {
    for (uint64_t i = blockIdx.x * (uint64_t)256 + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * 256) dst[i] = (uint32_t)i;
}

constexpr uint64_t SMALL = (uint64_t)24 << 20;   // nhood's half edge list at 1e6 spots
constexpr uint64_t LARGE = (uint64_t)2560 << 20; // a 160-batch label slab at 1e6 spots
constexpr int REREADS = 160;

__global__ __launch_bounds__(256) void k_calib_reread_24m(const uint4* __restrict__ src, uint32_t* __restrict__ out, uint64_t n16, int reps) {
    uint32_t acc = 0;
    for (int r = 0; r < reps; ++r) {
        // rotate the block -> slice map every pass: an XCD's 4 MiB L2 never sees the slice it read last time
        const uint64_t b = (blockIdx.x + (uint64_t)r * 97) % gridDim.x;
        for (uint64_t i = b * 256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256) {
            const uint4 v = src[i];
            acc += v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

__global__ __launch_bounds__(256) void k_calib_stream_2g5(const uint4* __restrict__ src, uint32_t* __restrict__ out, uint64_t n16) {
    uint32_t acc = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256) {
        const uint4 v = src[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    CHECK(hipSetDevice(0));
    uint8_t *a = nullptr, *b = nullptr;
    uint32_t* out = nullptr;
    CHECK(hipMalloc(&a, BYTES));
    CHECK(hipMalloc(&b, BYTES));
    CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(a, 1, BYTES));
    CHECK(hipMemset(b, 2, BYTES));
    CHECK(hipDeviceSynchronize());
    const int grid = 256 * 16;
    for (int rep = 0; rep < 2; ++rep) {  // every kernel twice; between two reads of `a` the other 512 MiB buffer passes through the caches
        hipLaunchKernelGGL(k_calib_read_b16, dim3(grid), dim3(256), 0, 0, reinterpret_cast<const uint4*>(a), out, BYTES / 16);
        hipLaunchKernelGGL(k_calib_write_row16, dim3(grid), dim3(256), 0, 0, reinterpret_cast<uint4*>(b), BYTES / 16);
        hipLaunchKernelGGL(k_calib_read_quadrow, dim3(grid), dim3(256), 0, 0, a, out, BYTES / 16);
        hipLaunchKernelGGL(k_calib_write_b4, dim3(grid), dim3(256), 0, 0, reinterpret_cast<uint32_t*>(b), BYTES / 4);
        hipLaunchKernelGGL(k_calib_read_pair8, dim3(grid), dim3(256), 0, 0, reinterpret_cast<const uint2*>(a), out, BYTES / 8);
        CHECK(hipDeviceSynchronize());
    }
    CHECK(hipGetLastError());
    // ---- Infinity-Cache probes
    uint8_t *small = nullptr, *large = nullptr;
    CHECK(hipMalloc(&small, SMALL));
    CHECK(hipMalloc(&large, LARGE));
    CHECK(hipMemset(small, 3, SMALL));
    CHECK(hipMemset(large, 4, LARGE));
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float ms_reread = 0.f, ms_stream = 0.f;
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_calib_stream_2g5, dim3(grid), dim3(256), 0, 0, reinterpret_cast<const uint4*>(large), out, LARGE / 16);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms_stream, e0, e1));
        CHECK(hipEventRecord(e0, 0));  // (the 2.5 GiB stream has just flushed the Infinity Cache: the first pass of the re-read is cold)
        hipLaunchKernelGGL(k_calib_reread_24m, dim3(grid), dim3(256), 0, 0, reinterpret_cast<const uint4*>(small), out, SMALL / 16, REREADS);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms_reread, e0, e1));
    }
    CHECK(hipGetLastError());
    printf("{\"bytes_per_launch\": %llu, \"reread_bytes\": %llu, \"reread_passes\": %d, \"reread_ms\": %.4f, \"reread_GBps\": %.1f, "
           "\"stream_bytes\": %llu, \"stream_ms\": %.4f, \"stream_GBps\": %.1f}\n",
           (unsigned long long)BYTES, (unsigned long long)SMALL, REREADS, ms_reread, SMALL * (double)REREADS / (ms_reread * 1e-3) / 1e9,
           (unsigned long long)LARGE, ms_stream, LARGE / (ms_stream * 1e-3) / 1e9);
    return 0;
}

// Issue-rate micro-benchmark for the instruction mixes of the nhood kernels (gfx950).
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_ops.bin tools/ubench_ops.hip && tools/ubench_ops.bin [out.json]
//
// Every kernel runs 8 independent dependency chains of ONE instruction per lane (32 instructions per loop trip), 8 waves
// per SIMD on every CU, and reports wave-instructions/s for the whole chip, the implied cycles per wave-instruction per
// SIMD at the nominal 2.4 GHz, and the ratio to v_fma_f32 (the guide's 2-cycle reference) measured in the same process —
// the ratio is what the kernel ceilings in bench.py use, it does not depend on the clock the chip actually sustains.
// LDS rows: ds_add_u32 (no return) per CU for the address patterns of the count kernel.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(x)                                                                          \
    do {                                                                                  \
        hipError_t e__ = (x);                                                             \
        if (e__ != hipSuccess) {                                                          \
            fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e__), __FILE__, __LINE__); \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

// ---- VALU rows: OP(d) expands to one instruction on accumulator %d with the loop-invariant operands %8 (b) and %9 (c)
#define REP8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define VALU_KERNEL(NAME, OP)                                                                                       \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t seed, int iters) {                          \
        uint32_t a0 = threadIdx.x * 2654435761u + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 7, a3 = a0 ^ 0x9e3779b9u,   \
                 a4 = a0 + 77, a5 = a1 * 9, a6 = a2 ^ a1, a7 = a3 + a2;                                             \
        uint32_t b = seed | 0x10001u, c = seed * 7 + 0x30003u;                                                      \
        for (int i = 0; i < iters; ++i) {                                                                           \
            asm volatile(REP8(OP) REP8(OP) REP8(OP) REP8(OP)                                                        \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)           \
                         : "v"(b), "v"(c));                                                                         \
        }                                                                                                           \
        out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                                \
    }

#define OP_FMA(d) "v_fma_f32 %" #d ", %" #d ", %8, %9\n"
#define OP_ADD(d) "v_add_u32 %" #d ", %" #d ", %8\n"
#define OP_SUB(d) "v_sub_u32 %" #d ", %" #d ", %8\n"
#define OP_XOR(d) "v_xor_b32 %" #d ", %" #d ", %8\n"
#define OP_AND(d) "v_and_b32 %" #d ", %" #d ", %8\n"
#define OP_SHR(d) "v_lshrrev_b32 %" #d ", 3, %" #d "\n"
#define OP_MIN(d) "v_min_u32 %" #d ", %" #d ", %8\n"
#define OP_MAD24(d) "v_mad_u32_u24 %" #d ", %" #d ", %8, %9\n"
#define OP_MUL24(d) "v_mul_u32_u24 %" #d ", %" #d ", %8\n"
#define OP_MULLO(d) "v_mul_lo_u32 %" #d ", %" #d ", %8\n"
#define OP_MULHI(d) "v_mul_hi_u32 %" #d ", %" #d ", %8\n"
#define OP_PKMAD(d) "v_pk_mad_u16 %" #d ", %" #d ", %8, %9\n"
#define OP_PKMUL(d) "v_pk_mul_lo_u16 %" #d ", %" #d ", %8\n"
#define OP_PKADD(d) "v_pk_add_u16 %" #d ", %" #d ", %8\n"
#define OP_PKSUB(d) "v_pk_sub_u16 %" #d ", %" #d ", %8\n"
#define OP_PKSHR(d) "v_pk_lshrrev_b16 %" #d ", %8, %" #d "\n"
#define OP_PKMIN(d) "v_pk_min_u16 %" #d ", %" #d ", %8\n"
#define OP_ALIGNBIT(d) "v_alignbit_b32 %" #d ", %" #d ", %8, %9\n"
#define OP_BFE(d) "v_bfe_u32 %" #d ", %" #d ", 8, 8\n"
#define OP_PERM(d) "v_perm_b32 %" #d ", %" #d ", %8, %9\n"
#define OP_LSHLOR(d) "v_lshl_or_b32 %" #d ", %" #d ", 8, %8\n"
#define OP_LSHLADD(d) "v_lshl_add_u32 %" #d ", %" #d ", 3, %8\n"
#define OP_XAD(d) "v_xad_u32 %" #d ", %" #d ", %8, %9\n"
#define OP_ANDOR(d) "v_and_or_b32 %" #d ", %" #d ", %8, %9\n"
#define OP_ADD3(d) "v_add3_u32 %" #d ", %" #d ", %8, %9\n"
#define OP_MIN3(d) "v_min3_u32 %" #d ", %" #d ", %8, %9\n"
#define OP_BFI(d) "v_bfi_b32 %" #d ", %8, %" #d ", %9\n"
#define OP_CNDMASK(d) "v_cndmask_b32 %" #d ", %" #d ", %8, vcc\n"
#define OP_CMPADDC(d) "v_cmp_ge_u32 vcc, %" #d ", %8\n v_addc_co_u32 %" #d ", vcc, %" #d ", 0, vcc\n"
#define OP_SDWA_MUL(d) \
    "v_mul_u32_u24_sdwa %" #d ", %" #d ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
#define OP_SDWA_ADD(d) \
    "v_add_u32_sdwa %" #d ", %" #d ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n"
#define OP_SDWA_W1(d) \
    "v_add_u32_sdwa %" #d ", %" #d ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n"
#define OP_DPP_QUAD(d) "v_mov_b32_dpp %" #d ", %" #d " quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n"
#define OP_DPP_ADD(d) "v_add_u32_dpp %" #d ", %" #d ", %8 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n"
#define OP_DOT2(d) "v_dot2_u32_u16 %" #d ", %" #d ", %8, %9\n"
#define OP_DOT4(d) "v_dot4_u32_u8 %" #d ", %" #d ", %8, %9\n"
#define OP_CVTUB(d) "v_cvt_f32_ubyte1 %" #d ", %" #d "\n"
#define OP_CVTU32(d) "v_cvt_u32_f32 %" #d ", %" #d "\n"
#define OP_MADU16(d) "v_mad_u32_u16 %" #d ", %" #d ", %8, %9\n"
#define OP_LSHL(d) "v_lshlrev_b32 %" #d ", 3, %" #d "\n"
#define OP_OR(d) "v_or_b32 %" #d ", %" #d ", %8\n"
#define OP_ADDLSHL(d) "v_add_lshl_u32 %" #d ", %" #d ", %8, 3\n"
#define OP_MAD64(d) "v_mad_u64_u32 v[20:21], vcc, %" #d ", %8, v[20:21]\n"

VALU_KERNEL(k_fma, OP_FMA)
VALU_KERNEL(k_add, OP_ADD)
VALU_KERNEL(k_sub, OP_SUB)
VALU_KERNEL(k_xor, OP_XOR)
VALU_KERNEL(k_and, OP_AND)
VALU_KERNEL(k_shr, OP_SHR)
VALU_KERNEL(k_min, OP_MIN)
VALU_KERNEL(k_mad24, OP_MAD24)
VALU_KERNEL(k_mul24, OP_MUL24)
VALU_KERNEL(k_mullo, OP_MULLO)
VALU_KERNEL(k_mulhi, OP_MULHI)
VALU_KERNEL(k_pkmad, OP_PKMAD)
VALU_KERNEL(k_pkmul, OP_PKMUL)
VALU_KERNEL(k_pkadd, OP_PKADD)
VALU_KERNEL(k_pksub, OP_PKSUB)
VALU_KERNEL(k_pkshr, OP_PKSHR)
VALU_KERNEL(k_pkmin, OP_PKMIN)
VALU_KERNEL(k_alignbit, OP_ALIGNBIT)
VALU_KERNEL(k_bfe, OP_BFE)
VALU_KERNEL(k_perm, OP_PERM)
VALU_KERNEL(k_lshlor, OP_LSHLOR)
VALU_KERNEL(k_lshladd, OP_LSHLADD)
VALU_KERNEL(k_xad, OP_XAD)
VALU_KERNEL(k_andor, OP_ANDOR)
VALU_KERNEL(k_add3, OP_ADD3)
VALU_KERNEL(k_min3, OP_MIN3)
VALU_KERNEL(k_bfi, OP_BFI)
VALU_KERNEL(k_cndmask, OP_CNDMASK)
VALU_KERNEL(k_cmpaddc, OP_CMPADDC)
VALU_KERNEL(k_sdwa_mul, OP_SDWA_MUL)
VALU_KERNEL(k_sdwa_add, OP_SDWA_ADD)
VALU_KERNEL(k_sdwa_w1, OP_SDWA_W1)
VALU_KERNEL(k_dpp_quad, OP_DPP_QUAD)
VALU_KERNEL(k_dpp_add, OP_DPP_ADD)
VALU_KERNEL(k_dot2, OP_DOT2)
VALU_KERNEL(k_dot4, OP_DOT4)
VALU_KERNEL(k_cvtub, OP_CVTUB)
VALU_KERNEL(k_cvtu32, OP_CVTU32)
VALU_KERNEL(k_madu16, OP_MADU16)
VALU_KERNEL(k_lshl, OP_LSHL)
VALU_KERNEL(k_or, OP_OR)
VALU_KERNEL(k_addlshl, OP_ADDLSHL)

// ---- LDS rows: 16 ds_add_u32 (no return) per trip on 8 address registers; MODE selects the address pattern
//   0: lane*4 (+ per-register offset)        conflict-free, the guide's ds_write_b32-class figure
//   1: pseudo-random words of a 57.6 KB table  (K=30, B=16 histogram addressed by random pairs)
//   2: all lanes of a quad on consecutive words of a random 64-byte pair row (the count kernel's actual pattern)
//   3: every lane the same address          worst case
template <int MODE>
__global__ __launch_bounds__(256) void k_lds_add(uint32_t* out, uint32_t seed, int iters, int words) {
    extern __shared__ uint32_t lds[];
    for (int i = threadIdx.x; i < words; i += 256) lds[i] = 0;
    __syncthreads();
    uint32_t ad[8];
    uint32_t h = threadIdx.x * 2654435761u + seed + blockIdx.x * 40503u;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        h = h * 1664525u + 1013904223u;
        uint32_t w;
        if (MODE == 0) w = (threadIdx.x + r * 256) % words;
        else if (MODE == 1) w = (h >> 8) % words;
        else if (MODE == 2) {
            uint32_t hq = ((threadIdx.x >> 2) * 2654435761u + seed + r * 97u) * 1664525u + 1013904223u;
            w = (((hq >> 8) % (words / 16)) * 16 + (threadIdx.x & 3) * 4 + ((r + (threadIdx.x >> 2)) & 3)) % words;
        } else w = r;
        ad[r] = w * 4;
    }
    uint32_t one = 1;
    for (int i = 0; i < iters; ++i) {
        asm volatile(
            "ds_add_u32 %0, %8\n ds_add_u32 %1, %8\n ds_add_u32 %2, %8\n ds_add_u32 %3, %8\n"
            "ds_add_u32 %4, %8\n ds_add_u32 %5, %8\n ds_add_u32 %6, %8\n ds_add_u32 %7, %8\n"
            "ds_add_u32 %0, %8 offset:4096\n ds_add_u32 %1, %8 offset:4096\n ds_add_u32 %2, %8 offset:4096\n"
            "ds_add_u32 %3, %8 offset:4096\n ds_add_u32 %4, %8 offset:4096\n ds_add_u32 %5, %8 offset:4096\n"
            "ds_add_u32 %6, %8 offset:4096\n ds_add_u32 %7, %8 offset:4096\n"
            "s_waitcnt lgkmcnt(0)\n"
            : "+v"(ad[0]), "+v"(ad[1]), "+v"(ad[2]), "+v"(ad[3]), "+v"(ad[4]), "+v"(ad[5]), "+v"(ad[6]), "+v"(ad[7])
            : "v"(one)
            : "memory");
    }
    __syncthreads();
    uint32_t s = 0;
    for (int i = threadIdx.x; i < words; i += 256) s += lds[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// the same with the address arithmetic of the count kernel in front of every atomic (2 SDWA + shift-add): shows whether
// the VALU work hides behind the LDS pipe
struct Row {
    std::string name;
    double winstr_per_s;
    double clk;  // per wave-instruction per SIMD (VALU) or per CU (LDS) at 2.4 GHz
    int per_trip;
};

int main(int argc, char** argv) {
    int dev = 0;
    CHECK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * 8;
    uint32_t* out = nullptr;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::vector<Row> rows;
    const double F = 2.4e9;

    auto time_valu = [&](const char* name, void (*kern)(uint32_t*, uint32_t, int), int per_trip) {
        const int iters = 4096;
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1u, 64);
        CHECK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1u + rep, iters);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double winstr = (double)blocks * 4 /*waves*/ * iters * per_trip;
        const double rate = winstr / (best * 1e-3);
        rows.push_back({name, rate, cus * 4 * F / rate, per_trip});
    };
#define TV(NAME, KERN) time_valu(NAME, KERN, 32)
    TV("v_fma_f32", k_fma);
    TV("v_add_u32", k_add);
    TV("v_sub_u32", k_sub);
    TV("v_xor_b32", k_xor);
    TV("v_and_b32", k_and);
    TV("v_lshrrev_b32", k_shr);
    TV("v_min_u32", k_min);
    TV("v_mad_u32_u24", k_mad24);
    TV("v_mul_u32_u24", k_mul24);
    TV("v_mul_lo_u32", k_mullo);
    TV("v_mul_hi_u32", k_mulhi);
    TV("v_pk_mad_u16", k_pkmad);
    TV("v_pk_mul_lo_u16", k_pkmul);
    TV("v_pk_add_u16", k_pkadd);
    TV("v_pk_sub_u16", k_pksub);
    TV("v_pk_lshrrev_b16", k_pkshr);
    TV("v_pk_min_u16", k_pkmin);
    TV("v_alignbit_b32", k_alignbit);
    TV("v_bfe_u32", k_bfe);
    TV("v_perm_b32", k_perm);
    TV("v_lshl_or_b32", k_lshlor);
    TV("v_lshl_add_u32", k_lshladd);
    TV("v_xad_u32", k_xad);
    TV("v_and_or_b32", k_andor);
    TV("v_add3_u32", k_add3);
    TV("v_min3_u32", k_min3);
    TV("v_bfi_b32", k_bfi);
    TV("v_cndmask_b32", k_cndmask);
    time_valu("v_cmp_ge_u32+v_addc_co_u32 (pair)", k_cmpaddc, 64);
    TV("v_mul_u32_u24_sdwa(BYTE_1)", k_sdwa_mul);
    TV("v_add_u32_sdwa(BYTE_2)", k_sdwa_add);
    TV("v_add_u32_sdwa(WORD_1)", k_sdwa_w1);
    TV("v_mov_b32_dpp(quad_perm)", k_dpp_quad);
    TV("v_add_u32_dpp(quad_perm)", k_dpp_add);
    TV("v_dot2_u32_u16", k_dot2);
    TV("v_dot4_u32_u8", k_dot4);
    TV("v_cvt_f32_ubyte1", k_cvtub);
    TV("v_cvt_u32_f32", k_cvtu32);
    TV("v_mad_u32_u16", k_madu16);
    TV("v_lshlrev_b32", k_lshl);
    TV("v_or_b32", k_or);
    TV("v_add_lshl_u32", k_addlshl);
    const double fma_rate = rows[0].winstr_per_s;

    std::vector<Row> lrows;
    auto time_lds = [&](const char* name, void (*kern)(uint32_t*, uint32_t, int, int), int words) {
        const int iters = 2048;
        const size_t lds = (size_t)words * 4;
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const int per_cu = (int)std::min<size_t>(8, (160 * 1024) / lds);
        const int nb = cus * per_cu;
        hipLaunchKernelGGL(kern, dim3(nb), dim3(256), lds, 0, out, 1u, 16, words);
        CHECK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(kern, dim3(nb), dim3(256), lds, 0, out, 1u + rep, iters, words);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double winstr = (double)nb * 4 * iters * 16;
        const double rate = winstr / (best * 1e-3);
        lrows.push_back({std::string(name) + " (" + std::to_string(per_cu * 4) + " waves/CU)", rate, cus * F / rate, 16});
    };
    time_lds("ds_add_u32 conflict-free lane*4, 16 KB/block", k_lds_add<0>, 4096);
    time_lds("ds_add_u32 random words of 57.6 KB", k_lds_add<1>, 14400);
    time_lds("ds_add_u32 count-kernel pattern (quad on one 64-B pair row, staggered), 57.6 KB", k_lds_add<2>, 14400);
    time_lds("ds_add_u32 count-kernel pattern, 29.8 KB (triangular pairs)", k_lds_add<2>, 7440);
    time_lds("ds_add_u32 all lanes one address", k_lds_add<3>, 4096);

    std::string js = "{\n  \"device\": \"" + std::string(prop.name) + "\", \"cus\": " + std::to_string(cus) +
                     ", \"nominal_clock_hz\": 2.4e9,\n  \"valu\": [\n";
    printf("%-40s %14s %10s %8s\n", "instruction", "wave-instr/s", "clk/SIMD", "vs fma");
    for (size_t i = 0; i < rows.size(); ++i) {
        const Row& r = rows[i];
        printf("%-40s %14.4g %10.2f %8.2f\n", r.name.c_str(), r.winstr_per_s, r.clk, fma_rate / r.winstr_per_s);
        char buf[512];
        snprintf(buf, sizeof(buf), "    {\"op\": \"%s\", \"wave_instr_per_s\": %.6g, \"clk_per_wave_instr_per_simd\": %.3f, \"cost_vs_v_fma_f32\": %.3f}%s\n",
                 r.name.c_str(), r.winstr_per_s, r.clk, fma_rate / r.winstr_per_s, i + 1 < rows.size() ? "," : "");
        js += buf;
    }
    js += "  ],\n  \"lds\": [\n";
    printf("%-90s %14s %10s\n", "LDS pattern", "wave-instr/s", "clk/CU");
    for (size_t i = 0; i < lrows.size(); ++i) {
        const Row& r = lrows[i];
        printf("%-90s %14.4g %10.2f\n", r.name.c_str(), r.winstr_per_s, r.clk);
        char buf[512];
        snprintf(buf, sizeof(buf), "    {\"pattern\": \"%s\", \"wave_instr_per_s\": %.6g, \"clk_per_wave_instr_per_cu\": %.3f, \"lane_atomics_per_s\": %.6g}%s\n",
                 r.name.c_str(), r.winstr_per_s, r.clk, r.winstr_per_s * 64, i + 1 < lrows.size() ? "," : "");
        js += buf;
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

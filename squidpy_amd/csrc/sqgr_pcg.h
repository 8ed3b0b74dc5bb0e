// libsqgr internal: numpy-compatible permutation streams on the device (PCG64 + Generator.shuffle / permutation).
// Implemented in sqgr_pcg.hip.
#pragma once
#include <vector>

#include "sqgr_common.h"

namespace sqgr {

struct PcgWorkspace {
    DevBuf<uint64_t> jump;  // [34][4] LCG jump-ahead table of the wave-per-permutation kernel
    DevBuf<uint8_t> rows;   // [permutation][n_pad] row-major shuffle workspace
    DevBuf<uint32_t> span;  // {0, n}: the single "library" of a plain permutation
    DevBuf<int32_t> cols;   // column workspace of the one-thread-per-permutation kernel (SQGR_PCG_KERNEL=lane)
    // the bucketed replay (long arrays): time-ordered swap records per (permutation, phase, 64-record block), their directory
    // (range | count << 8 | ordinal << 16 per block) and the blocks written per phase; library geometry cached from the device
    DevBuf<uint32_t> recs, dir, nblk, lib_phase;
    std::vector<uint32_t> lib_off_h;
    const uint32_t* lib_off_key = nullptr;
    int lib_phase_logs = -1, phases = 0;
};

// W[pos * stride + q] = element `pos` of the byte array numpy's generator q (states_dev row q: state_hi, state_lo, inc_hi,
// inc_lo) leaves behind after shuffling base_pos library by library (lib_off_dev: n_libs + 1 position offsets), q < pc.
int pcg_shuffle_labels(sqgr_ctx* ctx, PcgWorkspace& ws, int64_t n, int n_libs, const uint32_t* lib_off_dev, const uint8_t* base_pos_dev,
                       const uint64_t* states_dev, int64_t pc, int64_t stride, uint8_t* W, hipStream_t st, const char* timer_name);

// The same arrays as ROWS: ws.rows[q * (*row_stride) + pos] (no transposition; not for SQGR_PCG_KERNEL=lane: pcg_rows_available()).
bool pcg_rows_available();
int pcg_shuffle_rows(sqgr_ctx* ctx, PcgWorkspace& ws, int64_t n, int n_libs, const uint32_t* lib_off_dev, const uint8_t* base_pos_dev,
                     const uint64_t* states_dev, int64_t pc, hipStream_t st, const char* timer_name, int64_t* row_stride);

// idx_dev[q * n + i] = element i of `Generator.permutation(n)` drawn by generator q (states_dev row q), q < pc.
int pcg_permutations_dev(sqgr_ctx* ctx, PcgWorkspace& ws, int64_t n, const uint64_t* states_dev, int64_t pc, int32_t* idx_dev,
                         hipStream_t st);

}  // namespace sqgr

// libsqgr internal: the uniform cell list ("grid") of a 2-D point set, shared by the graph builders (sqgr_neighbors.hip)
// and Ripley's nearest-neighbour statistics (sqgr_ripley.hip).
#pragma once
#include "sqgr_common.h"

#include <vector>

namespace sqgr {

struct CellGrid {
    double x0, y0, inv_h, h;
    int gx, gy;
};

__device__ __forceinline__ void cell_of(const CellGrid& g, double x, double y, int& cx, int& cy) {
    cx = min(max((int)floor((x - g.x0) * g.inv_h), 0), g.gx - 1);
    cy = min(max((int)floor((y - g.y0) * g.inv_h), 0), g.gy - 1);
}

struct HostGrid {
    CellGrid g;
    std::vector<double> sx, sy;
    std::vector<int32_t> sid, cell_start;
};

// counting sort of the points into ~n / target_per_cell square cells of side >= min_h (sqgr_neighbors.hip)
int build_grid(const double* xy, int64_t n, double target_per_cell, double min_h, HostGrid& out);

struct DevGrid {
    DevBuf<double> sx, sy;
    DevBuf<int32_t> sid, cell_start;
    int upload(const HostGrid& h, hipStream_t st) {
        SQGR_TRY(sx.alloc(h.sx.size()));
        SQGR_TRY(sy.alloc(h.sy.size()));
        SQGR_TRY(sid.alloc(h.sid.size()));
        SQGR_TRY(cell_start.alloc(h.cell_start.size()));
        SQGR_HIP(hipMemcpyAsync(sx.p, h.sx.data(), h.sx.size() * 8, hipMemcpyHostToDevice, st));
        SQGR_HIP(hipMemcpyAsync(sy.p, h.sy.data(), h.sy.size() * 8, hipMemcpyHostToDevice, st));
        SQGR_HIP(hipMemcpyAsync(sid.p, h.sid.data(), h.sid.size() * 4, hipMemcpyHostToDevice, st));
        SQGR_HIP(hipMemcpyAsync(cell_start.p, h.cell_start.data(), h.cell_start.size() * 4, hipMemcpyHostToDevice, st));
        return SQGR_OK;
    }
};

}  // namespace sqgr

import sys, time, numpy as np
sys.path.insert(0, ".")
from squidpy_amd import _lib
ctx = _lib.default_context()
n, G = 100000, 20000
X = np.empty((n, G), dtype=np.float64); X[:] = 1.0
for it in range(2):
    t = time.perf_counter(); dm = _lib.DeviceMatrix(ctx, X); ctx.sync(); dt = time.perf_counter() - t; dm.close()
    print("whole 16 GB: %.3f s  %.1f GB/s" % (dt, X.nbytes / dt / 1e9))
for w in (2048, 512):
    for it in range(3):
        t = time.perf_counter(); dm = _lib.DeviceMatrix(ctx, X[:, 4096:4096 + w]); ctx.sync(); dt = time.perf_counter() - t; dm.close()
        print("block of %d columns (strided 2D): %.3f s  %.1f GB/s" % (w, dt, n * w * 8 / dt / 1e9))
X32 = X.astype(np.float32)
for it in range(2):
    t = time.perf_counter(); dm = _lib.DeviceMatrix(ctx, X32[:, 4096:4096 + 2048]); ctx.sync(); dt = time.perf_counter() - t; dm.close()
    print("f32 block of 2048 columns: %.3f s  %.1f GB/s" % (dt, n * 2048 * 4 / dt / 1e9))

"""How much page-locked host memory one process gets on this box: the largest single
allocation, and how many 16 GiB slabs can be held at once."""
import ctypes
import sys
import time

sys.path.insert(0, ".")
from snappy_amd import native

L = native.lib()
native.check(L.snappy_init(0), "init")
for gb in (64, 72, 80, 88):
    p = ctypes.c_void_p()
    t0 = time.perf_counter()
    st = L.snappy_host_alloc(gb << 30, ctypes.byref(p))
    print(f"single {gb} GiB: status {st}, {time.perf_counter() - t0:.2f} s", flush=True)
    if st != 0:
        break
    L.snappy_host_free(p)
held = []
for k in range(10):
    p = ctypes.c_void_p()
    t0 = time.perf_counter()
    st = L.snappy_host_alloc(16 << 30, ctypes.byref(p))
    print(f"slab {k} (total {16 * (k + 1)} GiB): status {st}, {time.perf_counter() - t0:.2f} s", flush=True)
    if st != 0:
        break
    held.append(p)
for p in held:
    L.snappy_host_free(p)

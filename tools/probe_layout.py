"""which local layout the distributed substructured solver gets for a few local sizes (GPU box)"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oldoceananigans_jl_amd as ocn
from oldoceananigans_jl_amd import _lib
arch = ocn.GPU(0)
for size in ((16, 16, 8), (32, 16, 8), (32, 32, 32), (64, 32, 16), (64, 64, 64), (128, 24, 16), (192, 8, 8), (256, 24, 16), (128, 128, 128), (64, 256, 256), (256, 256, 256)):
    grid = ocn.RectilinearGrid(arch, size=size, extent=(1, 1, 1), topology=(ocn.FullyConnected, ocn.Periodic, ocn.Periodic))
    h = C.c_void_p()
    rc = _lib.lib().ocn_dist_poisson_create(C.byref(h), grid.handle, 2, 0, 2.0)
    if rc:
        print(size, "create failed", rc); continue
    lay = C.c_int()
    _lib.lib().ocn_dist_poisson_layout(h, C.byref(lay))
    print(size, "layout", lay.value, flush=True)
    _lib.lib().ocn_dist_poisson_destroy(h)

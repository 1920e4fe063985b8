import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oldoceananigans_jl_amd as ocn
arch = ocn.GPU(0)
for strided in (0, 1):
    ocn.set_option("c2r_strided", strided)
    for size, topo, z in (((16,16,16), (ocn.Periodic,)*3, (0.0,1.0)), ((16,16,16), (ocn.Periodic,ocn.Periodic,ocn.Bounded), (-1.0,0.0)), ((32,32,32), (ocn.Periodic,)*3, (0.0,1.0)), ((256,256,256), (ocn.Periodic,)*3, (0.0,1.0))):
        grid = ocn.RectilinearGrid(arch, size=size, x=(0.0,1.0), y=(0.0,1.0), z=z, topology=topo)
        t = time.time()
        m = ocn.NonhydrostaticModel(grid=grid)
        print("strided", strided, size, topo[2].__name__, f"{time.time()-t:.2f}s", flush=True)
        del m

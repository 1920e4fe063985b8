import sys, ctypes as C
sys.path.insert(0,'/root/repo')
import oldoceananigans_jl_amd as ocn
from oldoceananigans_jl_amd import _lib
ocn.GPU(0)
for var in (1,2):
    for e in (-30,-27,-1,0,1,10,60):
        n=C.c_ulonglong()
        _lib.check(_lib.lib().ocn_debug_rcp_check(var,e,C.byref(n)))
        print(var,e,n.value)

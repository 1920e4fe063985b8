"""Architectures (reference: src/Architectures.jl:35-123, ext/OceananigansAMDGPUExt.jl:34-113).

`GPU()` here is the MI355X-native architecture: every kernel `launch!`ed on it is a hand-written HIP kernel behind the
C ABI. There is deliberately no `CPU()` in the product: the CPU path of this repository is the oracle under oracle/,
which only the tests may use."""
from . import _lib


class GPU:
    """`GPU(ROCBackend())` replacement. device_id follows `AC.device!(::ROCGPU, i)`."""

    def __init__(self, device_id=0):
        self.device_id = int(device_id)
        _lib.check(_lib.lib().ocn_init(self.device_id))

    def __repr__(self):
        return f"GPU{{MI355XNative}}(device={self.device_id})"


def ndevices():
    """ndevices(arch): HIP devices visible to this process"""
    import ctypes as C
    n = C.c_int()
    _lib.check(_lib.lib().ocn_device_count(C.byref(n)))
    return n.value


def architecture(obj):
    return obj.architecture


def synchronize(arch=None):
    """sync_device! (ext/OceananigansAMDGPUExt.jl:112-113)"""
    _lib.check(_lib.lib().ocn_sync())


def set_option(key, value):
    """library-wide tuning knobs (see include/ocn_mi355x.h: ocn_set_option)"""
    _lib.check(_lib.lib().ocn_set_option(key.encode(), int(value)))


def own_stream():
    """run on a stream created and owned by the library again (after distributed.init_process_group pointed it at torch's)"""
    _lib.check(_lib.lib().ocn_own_stream())

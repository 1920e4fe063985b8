"""Fields (reference: src/Fields/field.jl:23-38, src/Grids/new_data.jl:36-73).

A Field owns (or borrows) a dense device array WITH halos in the reference's parent layout: column-major, x fastest,
interior element (i, j, k) at (i-1+Hx, j-1+Hy, k-1+Hz)."""
import ctypes as C

import numpy as np

from . import _lib
from .grids import Bounded, Center, Face

_LOC_CODE = {Center: 0, Face: 1}


class Field:
    def __init__(self, loc, grid, data=None, owner=None):
        self.grid = grid
        self.loc = tuple(loc)
        self.shape = grid.total_size(self.loc)
        self.nbytes = int(np.prod(self.shape)) * 8
        self._owner = None   # views into a model's memory do NOT keep the model alive (no reference cycles)
        if data is None:
            p = C.c_void_p()
            _lib.check(_lib.lib().ocn_malloc(C.byref(p), self.nbytes))   # zeros(arch, FT, sz...)
            self.data = p
            self._owns = True
        else:
            self.data = data if isinstance(data, C.c_void_p) else C.c_void_p(data)
            self._owns = False

    @property
    def architecture(self):
        return self.grid.architecture

    @property
    def loc_codes(self):
        return tuple(_LOC_CODE[l] for l in self.loc)

    # ---- host transfers ------------------------------------------------------------------------------------------
    def parent(self):
        """Array(parent(field)): host copy of the whole haloed array (Fortran order)."""
        a = np.empty(self.shape, dtype=np.float64, order="F")
        _lib.check(_lib.lib().ocn_memcpy_d2h(a.ctypes.data, self.data, self.nbytes))
        return a

    def set_parent(self, a):
        a = np.asfortranarray(a, dtype=np.float64)
        if a.shape != self.shape:
            raise ValueError(f"parent shape {a.shape} != {self.shape}")
        _lib.check(_lib.lib().ocn_memcpy_h2d(self.data, a.ctypes.data, self.nbytes))

    def _interior_slices(self):
        g = self.grid
        return tuple(slice(h, h + n) for h, n in zip(g.halo_size, g.interior_size(self.loc)))

    def interior(self):
        """Array(interior(field)) (field.jl: interior)"""
        return self.parent()[self._interior_slices()]

    def set(self, value):
        """set!(field, value): value is an array of interior size, a number, or f(x, y, z)."""
        a = self.parent()
        view = a[self._interior_slices()]
        if callable(value):
            x, y, z = self.grid.nodes(self.loc)
            view[...] = value(x, y, z)
        else:
            view[...] = value
        self.set_parent(a)
        return self

    def __del__(self):
        if getattr(self, "_owns", False):
            try:
                _lib.lib().ocn_free(self.data)
            except Exception:
                pass


def CenterField(grid):
    return Field((Center, Center, Center), grid)


def XFaceField(grid):
    return Field((Face, Center, Center), grid)


def YFaceField(grid):
    return Field((Center, Face, Center), grid)


def ZFaceField(grid):
    return Field((Center, Center, Face), grid)


def interior(field):
    return field.interior()


def set_(field, value):
    return field.set(value)


def _ptr_array(fields):
    return (C.c_void_p * len(fields))(*[f.data for f in fields])


def _loc_array(fields):
    arr = ((C.c_int * 3) * len(fields))()
    for n, f in enumerate(fields):
        for d in range(3):
            arr[n][d] = f.loc_codes[d]
    return arr


def fill_halo_regions(fields, fill_open_bcs=True, boundary_conditions=None):
    """fill_halo_regions!(field | tuple of fields; fill_open_bcs) (BoundaryConditions/fill_halo_regions.jl:25-36) with the
    default boundary conditions (field_boundary_conditions.jl:15-25) or, per field, a FieldBoundaryConditions of constant
    Flux / Value / Gradient / Open conditions (`boundary_conditions`: one entry or None per field)."""
    if isinstance(fields, Field):
        fields = [fields]
        if boundary_conditions is not None and not isinstance(boundary_conditions, (list, tuple)):
            boundary_conditions = [boundary_conditions]
    fields = list(fields)
    if not fields:
        return
    grid = fields[0].grid
    if boundary_conditions is None:
        _lib.check(_lib.lib().ocn_fill_halo_regions(grid.handle, _ptr_array(fields), _loc_array(fields), len(fields),
                                                    int(fill_open_bcs)))
        return
    from .boundary_conditions import bc_table
    if len(boundary_conditions) != len(fields):
        raise ValueError("one FieldBoundaryConditions (or None) per field")
    _lib.check(_lib.lib().ocn_fill_halo_regions_bcs(grid.handle, _ptr_array(fields), _loc_array(fields), len(fields),
                                                    bc_table(list(boundary_conditions), grid), int(fill_open_bcs)))

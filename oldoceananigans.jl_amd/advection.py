"""Advection scheme descriptors (reference: src/Advection/weno_reconstruction.jl:77-93)."""


class WENO:
    """WENO(order=5): WENO{3, Float64, Float32} with buffer_scheme WENO{2} -> UpwindBiased{1} and
    advecting_velocity_scheme Centered(order=4). Only order 5 is accelerated."""

    def __init__(self, order=5, bounds=None):
        if order % 2 == 0:
            raise ValueError("WENO reconstruction scheme is defined only for odd orders")
        if order != 5 or bounds is not None:
            raise NotImplementedError("only WENO(order=5) without bounds is on the accelerated hot path")
        self.order = order

    def __repr__(self):
        return ("WENO{3, Float64, Float32}(order=5)\n├── buffer_scheme: WENO{2, Float64, Float32}(order=3)\n"
                "└── advection_velocity_scheme: Centered(order=4)")

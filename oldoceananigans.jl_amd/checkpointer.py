"""Checkpoint / restore of a model state (reference: src/OutputWriters/checkpointer.jl:161-231, output_writer_utils.jl:155-170).

The reference writes JLD2 (an HDF5 dialect); no HDF5 library exists in this environment and the reference ships no `.jld2` file to
check a hand-written one against, so the CONTAINER here is NumPy's `.npz` -- **not readable by Oceananigans** (SURVEY.md 8f.3 stays
open) -- while the CONTENT follows the reference's layout: one entry per address the reference's `write_output!` creates,

    NonhydrostaticModel/<field>/data                     parent array of every prognostic field, halos included
    NonhydrostaticModel/timestepper/Gⁿ/<field>/data      (serializeproperty!(::RungeKutta3TimeStepper | ::QuasiAdamsBashforth2TimeStepper))
    NonhydrostaticModel/timestepper/G⁻/<field>/data
    NonhydrostaticModel/clock/{time, iteration, stage, last_Δt, last_stage_Δt}
    NonhydrostaticModel/grid/{size, halo}                (validated on restore)

`set_from_checkpoint` is `set!(model, filepath)`: copy the arrays, set the clock; the next time-step continues bit for bit where the
checkpointed run would have gone (tests/test_gpu_parity.py::test_checkpoint_and_restore_continue_bit_identically).
One deliberate difference in the content: this library's Gⁿ holds the Flux-boundary-condition terms between time-steps,
the reference adds them right before each substep -- a model without valued Flux conditions writes the same Gⁿ."""
import ctypes as C

import numpy as np

from . import _lib

ADDRESS = "NonhydrostaticModel"          # checkpointer_address(::NonhydrostaticModel)


def write_checkpoint(model, filepath):
    """write_output!(::Checkpointer, model) (checkpointer.jl:161-183) into `filepath` (.npz)"""
    out = {}
    for name, field in model.fields().items():
        out[f"{ADDRESS}/{name}/data"] = field.parent()
        out[f"{ADDRESS}/timestepper/Gⁿ/{name}/data"] = model.tendency(name).parent()
        out[f"{ADDRESS}/timestepper/G⁻/{name}/data"] = model.tendency(name, previous=True).parent()
    clk = model.clock
    out[f"{ADDRESS}/clock/time"] = np.float64(clk.time)
    out[f"{ADDRESS}/clock/iteration"] = np.int64(clk.iteration)
    out[f"{ADDRESS}/clock/stage"] = np.int64(clk.stage)
    out[f"{ADDRESS}/clock/last_Δt"] = np.float64(clk.last_Δt)
    out[f"{ADDRESS}/clock/last_stage_Δt"] = np.float64(clk.last_stage_Δt)
    grid = model.grid.local if hasattr(model.grid, "local") else model.grid
    out[f"{ADDRESS}/grid/size"] = np.asarray(grid.size, dtype=np.int64)
    out[f"{ADDRESS}/grid/halo"] = np.asarray(grid.halo_size, dtype=np.int64)
    np.savez(filepath, **out)
    return filepath if str(filepath).endswith(".npz") else str(filepath) + ".npz"


def set_from_checkpoint(model, filepath):
    """set!(model, filepath::AbstractString) (checkpointer.jl:199-231): prognostic fields, tendencies and clock from the file"""
    from .models import update_state
    with np.load(filepath, allow_pickle=False) as file:
        grid = model.grid.local if hasattr(model.grid, "local") else model.grid
        if tuple(file[f"{ADDRESS}/grid/size"]) != tuple(grid.size) or tuple(file[f"{ADDRESS}/grid/halo"]) != tuple(grid.halo_size):
            raise ValueError("the checkpointed grid does not match the model's grid")
        for name, field in model.fields().items():
            key = f"{ADDRESS}/{name}/data"
            if key not in file:
                import warnings
                warnings.warn(f"Field {name} does not exist in checkpoint and could not be restored.")
                continue
            field.set_parent(file[key])
            model.tendency(name).set_parent(file[f"{ADDRESS}/timestepper/Gⁿ/{name}/data"])
            model.tendency(name, previous=True).set_parent(file[f"{ADDRESS}/timestepper/G⁻/{name}/data"])
        _lib.check(_lib.lib().ocn_model_set_clock(model.handle, float(file[f"{ADDRESS}/clock/time"]), int(file[f"{ADDRESS}/clock/iteration"]),
                                                  int(file[f"{ADDRESS}/clock/stage"]), float(file[f"{ADDRESS}/clock/last_Δt"]),
                                                  float(file[f"{ADDRESS}/clock/last_stage_Δt"])))
    # auxiliary state (halos, eddy diffusivities, hydrostatic pressure, the tendencies themselves) follows from the prognostic fields
    update_state(model, True)

"""mitsuba-im_amd: MI355X (gfx950) wavefront path tracer behind Mitsuba-IM's Integrator boundary.

Python here is plumbing only (ctypes over the C-ABI in include/mi355pt.h, scene generators, torch.distributed glue);
the product is libmi355pt.so (mitsuba-im_amd/csrc).  The package directory name contains a hyphen, import it with
`importlib.import_module("mitsuba-im_amd")`.
"""
from . import scenes  # noqa: F401
from .api import (Lib, Scene, Render, MiError, lib, build, load_sobol_tables, device_sincosf, device_libm)  # noqa: F401

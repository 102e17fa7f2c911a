"""smvs_amd -- MI355X-native depth-optimisation hot path of smvs.

The product is csrc/libsmvs_hip.so behind the C ABI of include/smvs_hip.h;
this package is the thin host-side handle used by tests and bench.py.
"""
from . import _capi  # noqa: F401
from .device import (ViewContext, device_count, sgm_run, bilateral_upsample,  # noqa: F401
                     sgm_depth_for_view, cut_depth_maps)

__all__ = ["ViewContext", "device_count", "sgm_run", "bilateral_upsample",
           "sgm_depth_for_view", "cut_depth_maps"]

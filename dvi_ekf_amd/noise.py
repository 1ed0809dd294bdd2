"""Noise for the synthetic batch workloads: standard normals that are a pure
function of (seed, instance, step, lane) -- Philox4x32-10 with key = seed and counter =
(instance, step, lane), Box-Muller -- generated on the device by ``dvi_eskf_randn``
(csrc/eskf_noise.hip).  Instance i of a sharded run draws what it draws in the unsharded run."""
from __future__ import annotations

import ctypes as C

import torch

from . import _capi

IC_STEP = 0xFFFFFFFF        # the step word used for the draw of the initial condition


def _launch(kind, out, seed, first_instance, n_instances, first_step, n_steps, lanes):
    dev = out.device
    if dev.type != "cuda":
        raise RuntimeError("dvi_ekf_amd.noise generates on a HIP device only")
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    _capi.check(_capi.lib().dvi_eskf_randn(kind, dev.index if dev.index is not None else torch.cuda.current_device(),
                                           C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), int(first_instance),
                                           int(n_instances), int(first_step), int(n_steps), int(lanes),
                                           C.c_void_p(out.data_ptr()), stream), "dvi_eskf_randn")
    return out


def randn(seed: int, first_instance: int, n_instances: int, first_step: int, n_steps: int, lanes: int,
          dtype=torch.float64, device="cuda") -> torch.Tensor:
    """[n_steps, n_instances, lanes] standard normals for instances first_instance.. and steps first_step.."""
    if dtype not in (torch.float64, torch.float32):
        raise ValueError("dtype must be float64 or float32")
    out = torch.empty((n_steps, n_instances, lanes), dtype=dtype, device=torch.device(device))
    return _launch(0 if dtype == torch.float64 else 1, out, seed, first_instance, n_instances, first_step, n_steps, lanes)


def philox_bits(seed: int, first_instance: int, n_instances: int, first_step: int, n_steps: int, lanes: int,
                device="cuda") -> torch.Tensor:
    """The raw Philox4x32-10 words behind ``randn``: int32 view of uint32 [n_steps, n_instances, lanes, 4]."""
    out = torch.empty((n_steps, n_instances, lanes, 4), dtype=torch.int32, device=torch.device(device))
    return _launch(2, out, seed, first_instance, n_instances, first_step, n_steps, lanes)

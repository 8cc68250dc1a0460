"""`--amp 1` of the reference (autocast + torch.cuda.amp.GradScaler: /root/reference/code/flower_pCE_2D.py:47-48,104,
143-146; flower_common.py:466-468,576-584), MI355X-native.

* ``autocast(enabled=amp)`` maps to the reduced-precision compute mode of the HIP path: 16-bit storage / MFMA operands
  with fp32 accumulation, fp32 statistics, losses, master weights and optimizer.  The default is **fp16**
  (`set_compute_dtype(model, "fp16")`, FI_F16 kernels), the dtype the reference's ``torch.cuda.amp.autocast`` uses: its
  5 exponent bits are why the scaler below exists -- a scaled gradient that leaves fp16's range becomes inf, the step
  is skipped and the scale halved.  ``args.amp_dtype = "bf16"`` selects bf16 instead (same MFMA rate on CDNA4, fp32's
  exponent range: the scaler then never fires); bench.py's performance mode is bf16 without a scaler.
* ``GradScaler`` keeps the reference's control flow and state machine exactly (scale(loss).backward(); step(optimizer);
  update(): init_scale 2**16, x2 after 2000 clean steps, x0.5 and a SKIPPED optimizer step when a gradient is inf/NaN),
  with scale / growth tracker / found-inf flag resident on the device, so the whole iteration stays hipGraph-capturable.
"""
from __future__ import annotations

import torch

from . import _lib as L


class GradScaler:
    def __init__(self, init_scale=2.0 ** 16, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000, enabled=True,
                 device=None):
        self._enabled = enabled
        self._init_scale, self._growth, self._backoff, self._interval = float(init_scale), float(growth_factor), \
            float(backoff_factor), int(growth_interval)
        self._device = device
        self._scale = self._tracker = self._found_inf = None

    def _lazy(self, device):
        if self._scale is None:
            self._scale = torch.full((1,), self._init_scale, dtype=torch.float32, device=device)
            self._tracker = torch.zeros(1, dtype=torch.int32, device=device)
            self._found_inf = torch.zeros(1, dtype=torch.float32, device=device)

    def is_enabled(self):
        return self._enabled

    def scale(self, outputs):
        if not self._enabled:
            return outputs
        self._lazy(outputs.device)
        return outputs * self._scale.to(outputs.dtype).reshape(())

    def step(self, optimizer, *args, **kwargs):
        """Unscales the optimizer's gradients, then steps unless they hold an inf/NaN (GradScaler.step)."""
        if not self._enabled:
            return optimizer.step(*args, **kwargs)
        if not hasattr(optimizer, "step_scaled"):
            raise TypeError("fedicra_amd.amp.GradScaler drives fedicra_amd.optim.FusedAdamW")
        self._lazy(optimizer.model.flat_params.device)
        return optimizer.step_scaled(self._scale, self._found_inf)

    def unscale_range(self, grads):
        """scaler.unscale_ for one flat gradient range (an optimizer that does not step: the ALA loop's AdamW has lr = 0,
        flower_common.py:560,576-584): grads /= scale in place, the found-inf flag raised on inf/NaN.  Returns the flag
        (device fp32[1]) for kernels that must not consume an overflowed gradient."""
        self._lazy(grads.device)
        L.amp_unscale(grads, self._scale, self._found_inf)
        return self._found_inf

    def update(self, new_scale=None):
        if not self._enabled or self._scale is None:
            return
        if new_scale is not None:
            self._scale.fill_(float(new_scale))
            self._found_inf.zero_()
            return
        L.amp_update(self._scale, self._tracker, self._found_inf, self._growth, self._backoff, self._interval)

    def get_scale(self):
        if not self._enabled:
            return 1.0
        return self._init_scale if self._scale is None else float(self._scale.item())

    def state_dict(self):
        if not self._enabled or self._scale is None:
            return {}
        return {"scale": self.get_scale(), "growth_factor": self._growth, "backoff_factor": self._backoff,
                "growth_interval": self._interval, "_growth_tracker": int(self._tracker.item())}

    def load_state_dict(self, sd):
        if not sd:
            return
        dev = self._device or "cuda"
        self._lazy(dev)
        self._scale.fill_(float(sd["scale"]))
        self._tracker.fill_(int(sd["_growth_tracker"]))
        self._growth, self._backoff, self._interval = sd["growth_factor"], sd["backoff_factor"], sd["growth_interval"]


class autocast:
    """Context manager kept for call-site compatibility (`with autocast(enabled=self.amp)`): the compute dtype of the
    HIP modules is a property of the model (set_compute_dtype), so entering it changes nothing."""

    def __init__(self, enabled=True, **_):
        self.enabled = enabled

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

"""Loading reference-format checkpoints into the MI355X denoiser.

The reference's trainer saves ``{"state_dict", "optimizer", "lr_scheduler", "iter", "ema_state_dict"}``
(main.py:278-288) where ``ema_state_dict`` is an ``EMAModel.state_dict()`` whose ``shadow_params`` is a
LIST of tensors in ``model.parameters()`` order; its demo restores those EMA weights by position
(``copy_parameters``, app.py:15-19, 31-32).  ``DhariwalUNet`` registers its parameters under the
reference's names and in the reference's order, so both forms load here.

Files are read with ``torch.load(..., weights_only=True)``: nothing in the file is executed.
"""
from __future__ import annotations

from typing import Dict, Iterable, Union

import torch

from .model import Denoiser, DhariwalUNet


def _unet(model: Union[Denoiser, DhariwalUNet]) -> DhariwalUNet:
    return model.unet if isinstance(model, Denoiser) else model


def copy_parameters(from_parameters: Iterable[torch.Tensor], model: Union[Denoiser, DhariwalUNet]) -> None:
    """Positional copy of a parameter list into the model (app.py:15-19).  Lengths and shapes must agree."""
    net = _unet(model)
    to_parameters = list(net.parameters())
    from_parameters = list(from_parameters)
    if len(from_parameters) != len(to_parameters):
        raise AssertionError(f"checkpoint holds {len(from_parameters)} parameters, the model {len(to_parameters)}")
    names = [n for n, _ in net.named_parameters()]
    with torch.no_grad():
        for name, src, dst in zip(names, from_parameters, to_parameters):
            if tuple(src.shape) != tuple(dst.shape):
                raise ValueError(f"parameter {name}: checkpoint shape {tuple(src.shape)} != model {tuple(dst.shape)}")
            dst.data.copy_(src.to(device=dst.device, dtype=dst.dtype))
    net.invalidate()


def load_checkpoint(model: Union[Denoiser, DhariwalUNet], path: str, use_ema: bool = True,
                    strict: bool = True) -> Dict[str, object]:
    """Load a reference-format ``.pth``: EMA weights by position when present and ``use_ema``
    (what app.py does), otherwise ``state_dict`` by name (what main.py resumes from).  A bare
    ``state_dict`` file is accepted too.  Returns ``{"source": ..., "iter": ...}``."""
    state = torch.load(path, map_location="cpu", weights_only=True)
    if not isinstance(state, dict):
        raise ValueError(f"{path}: expected a dict checkpoint, got {type(state).__name__}")
    ema = state.get("ema_state_dict")
    if use_ema and isinstance(ema, dict) and "shadow_params" in ema:
        copy_parameters(ema["shadow_params"], model)
        return {"source": "ema_state_dict.shadow_params", "iter": state.get("iter")}
    sd = state.get("state_dict", state)
    _unet(model).load_state_dict(sd, strict=strict)
    return {"source": "state_dict", "iter": state.get("iter") if "state_dict" in state else None}

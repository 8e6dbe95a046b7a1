"""mad_amd -- MI355X-native denoising path of MAD (cross-domain diffusion for makeup tasks).

Public surface (names follow the reference's ``modeling`` package):
``build_model``, ``DhariwalUNet``, ``TranslationDiffusion``, ``CustomDDPMScheduler``,
``CustomDDIMScheduler``, ``create_cfg`` / ``merge_possible_with_base`` / ``load_cfg``.
Everything that computes loads ``libmad_hip.so``; there is no CPU fallback.
"""
__version__ = "0.1.0"

_LAZY = {
    "build_model": "model", "DhariwalUNet": "model", "Denoiser": "model",
    "TranslationDiffusion": "translation",
    "CustomDDPMScheduler": "scheduler", "CustomDDIMScheduler": "scheduler",
    "create_cfg": "config", "merge_possible_with_base": "config", "load_cfg": "config",
    "spec_from_cfg": "unet_spec", "seeded_state_dict": "init",
    "load_checkpoint": "checkpoint", "copy_parameters": "checkpoint",
}


def __getattr__(name):
    if name in _LAZY:
        import importlib
        return getattr(importlib.import_module(f".{_LAZY[name]}", __name__), name)
    raise AttributeError(name)


__all__ = list(_LAZY)

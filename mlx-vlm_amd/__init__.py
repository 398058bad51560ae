"""mlx-vlm_amd: MI355X-native engine for mlx-vlm's generate hot path.

Same Python surface as the reference for this path (mlx_vlm/__init__.py:7-21,
mlx_vlm/generate/__init__.py:57-104): load / generate / stream_generate /
batch_generate / generate_step, the `models.<model_type>` module contract and
GenerationResult.  All arithmetic runs in libvlm_hip.so (hand-written gfx950
kernels); importing the compute path without that library raises.
"""
__version__ = "0.1.0"

_LAZY = {
    "load": ("utils", "load"),
    "load_model": ("utils", "load_model"),
    "prepare_inputs": ("utils", "prepare_inputs"),
    "StoppingCriteria": ("utils", "StoppingCriteria"),
    "generate": ("generate", "generate"),
    "stream_generate": ("generate", "stream_generate"),
    "batch_generate": ("generate", "batch_generate"),
    "generate_step": ("generate", "generate_step"),
    "GenerationResult": ("generate", "GenerationResult"),
    "BatchResponse": ("generate", "BatchResponse"),
    "BatchGenerator": ("batch", "BatchGenerator"),
    "make_sampler": ("sample_utils", "make_sampler"),
    "apply_chat_template": ("prompt_utils", "apply_chat_template"),
    "get_message_json": ("prompt_utils", "get_message_json"),
    "BatchStats": ("generate", "BatchStats"),
    "PromptCacheState": ("generate", "PromptCacheState"),
    "VisionFeatureCache": ("vision_cache", "VisionFeatureCache"),
}


def __getattr__(name):
    if name in _LAZY:
        import importlib

        mod, attr = _LAZY[name]
        return getattr(importlib.import_module(f"{__name__}.{mod}"), attr)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")

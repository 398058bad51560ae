"""`models.llava_bunny` (nanoLLaVA): the reference's module contract for this model type
(mlx_vlm/models/llava_bunny/__init__.py)."""
from .config import ModelConfig, TextConfig, VisionConfig
from .language import LanguageModel
from .llava_bunny import Model, sanitize_keys
from .processing import ImageProcessor, assemble_input_ids
from .vision import VisionModel

__all__ = ["Model", "ModelConfig", "TextConfig", "VisionConfig", "LanguageModel", "VisionModel", "ImageProcessor",
           "assemble_input_ids", "sanitize_keys"]

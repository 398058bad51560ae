"""`models.phi3_v` (Phi-3.5-vision): the reference's module contract for this model type
(mlx_vlm/models/phi3_v/__init__.py)."""
from .config import ModelConfig, TextConfig, VisionConfig
from .language import LanguageModel
from .phi3_v import Model, sanitize_keys
from .processing_phi3_v import Phi3VImageProcessor, Phi3VProcessor
from .vision import VisionModel

__all__ = ["Model", "ModelConfig", "TextConfig", "VisionConfig", "LanguageModel", "VisionModel", "Phi3VImageProcessor",
           "Phi3VProcessor", "sanitize_keys"]

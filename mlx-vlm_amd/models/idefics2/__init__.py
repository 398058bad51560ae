"""`models.idefics2`: the reference's module contract for this model type (mlx_vlm/models/idefics2/__init__.py)."""
from .config import ModelConfig, PerceiverConfig, TextConfig, VisionConfig
from .connector import Connector
from .idefics2 import Model, sanitize_keys
from .language import LanguageModel
from .processing_idefics2 import Idefics2ImageProcessor, Idefics2Processor
from .vision import VisionModel

__all__ = ["Model", "ModelConfig", "TextConfig", "VisionConfig", "PerceiverConfig", "LanguageModel", "VisionModel", "Connector",
           "Idefics2ImageProcessor", "Idefics2Processor", "sanitize_keys"]

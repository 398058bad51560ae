from .config import ModelConfig, TextConfig, VisionConfig
from .language import LanguageModel
from .qwen2_vl import Model
from .vision import VisionModel

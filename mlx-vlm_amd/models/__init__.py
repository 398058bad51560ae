"""Model zoo with the reference's module contract (mlx_vlm/models/<model_type>/):
each package exports ModelConfig, TextConfig, VisionConfig, Model, LanguageModel,
VisionModel (reference mlx_vlm/models/qwen2_vl/__init__.py:1-4)."""

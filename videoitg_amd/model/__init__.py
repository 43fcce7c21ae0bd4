from .grounding_qwen2 import EagleQwenG, EagleQwenGConfig  # noqa: F401

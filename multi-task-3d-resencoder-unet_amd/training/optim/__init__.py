from .engine_adamw import EngineAdamW  # noqa: F401

from .model import WanModel  # noqa: F401

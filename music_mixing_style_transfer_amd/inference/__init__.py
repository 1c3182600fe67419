from . import segmentation  # noqa: F401
from .engine import StyleTransferEngine, build_models, embedding_mean  # noqa: F401

from .common_audioeffects import *  # noqa: F401,F403
from .common_audioeffects import (AugmentationChain, Compressor, Equaliser, Gain, MidSideImager, Parameter,
                                  ParameterList, Processor, rms_normalize_)

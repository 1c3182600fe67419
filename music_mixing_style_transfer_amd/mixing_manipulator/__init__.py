from .common_audioeffects import *  # noqa: F401,F403
from .common_audioeffects import (AlgorithmicReverb, AugmentationChain, Compressor, ConvolutionalReverb, Equaliser, Gain, Haas, MidSideImager,
                                  Panner, Parameter, ParameterList, Processor, rms_normalize_)
from .audio_effects_chain import (create_effects_augmentation_chain, create_inst_effects_augmentation_chain,  # noqa: F401
                                  load_impulse_responses)

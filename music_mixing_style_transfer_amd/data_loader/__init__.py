from .data_loader import Song_Dataset_Inference  # noqa: F401
from .loader_utils import (load_wav_length, SlicedWavWriter, load_wav_device, load_wav_segment, pcm16_device,  # noqa: F401
                           save_wav_pcm16)

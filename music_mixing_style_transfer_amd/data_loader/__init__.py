from .data_loader import Song_Dataset_Inference  # noqa: F401
from .loader_utils import load_wav_length, SlicedWavWriter, load_wav_segment, save_wav_pcm16  # noqa: F401

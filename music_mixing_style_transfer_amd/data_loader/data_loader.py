"""Inference dataset (reference data_loader/data_loader.py:545-603, Song_Dataset_Inference): for every song
directory <target_dir>/<song>/ load the 4 stems of the input and of the reference track(s) from
<song>/<stem_level_directory_name>[/<separation_model>]/<input|reference>/<stem>.wav, clamp to [-1, 1] and stack to
[4, 2, L] float32 tensors.

The FX normaliser applied to the INPUT stems when args.normalize_input is set (Audio_Effects_Normalizer,
mixing_manipulator/data_normalization.py) depends on pyloudnorm / librosa / aubio arithmetic that is not
reachable offline; it is a "next" row of the scope table, so normalize_input=True raises here.
"""
import os
from glob import glob

import torch

from .loader_utils import load_wav_segment


class Song_Dataset_Inference:
    def __init__(self, args):
        self.args = args
        self.data_dir = args.target_dir
        self.interpolate = args.interpolation
        self.instruments = args.instruments
        self.data_dir_paths = sorted(glob(f"{self.data_dir}*/"))
        self.input_name = args.input_file_name
        self.reference_name = args.reference_file_name
        self.stem_level_directory_name = args.stem_level_directory_name if args.do_not_separate \
            else os.path.join(args.stem_level_directory_name, args.separation_model)
        if self.interpolate:
            self.reference_name_B = args.reference_file_name_2interpolate
        if args.normalize_input:
            raise NotImplementedError(
                "normalize_input=True needs the Audio_Effects_Normalizer (pyloudnorm / librosa / aubio arithmetic), "
                "which is outside the accelerated hot path of this build; pass --normalize_input False and feed "
                "pre-normalised stems")

    def __len__(self):
        return len(self.data_dir_paths)

    def _stem(self, idx, which, inst):
        path = os.path.join(self.data_dir_paths[idx], self.stem_level_directory_name, which, inst + ".wav")
        wav = load_wav_segment(path, axis=0, sample_rate=self.args.sample_rate)
        return torch.clamp(torch.from_numpy(wav).float(), min=-1, max=1)

    def __getitem__(self, idx):
        inputs = [self._stem(idx, self.input_name, i) for i in self.instruments]
        refs = [self._stem(idx, self.reference_name, i) for i in self.instruments]
        dir_name = os.path.dirname(self.data_dir_paths[idx])
        if self.interpolate:
            refs_b = [self._stem(idx, self.reference_name_B, i) for i in self.instruments]
            return torch.stack(inputs, 0), torch.stack(refs, 0), torch.stack(refs_b, 0), dir_name
        return torch.stack(inputs, 0), torch.stack(refs, 0), dir_name

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

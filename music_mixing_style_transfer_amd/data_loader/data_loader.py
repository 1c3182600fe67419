"""Inference dataset (reference data_loader/data_loader.py:545-603, Song_Dataset_Inference): for every song
directory <target_dir>/<song>/ load the 4 stems of the input and of the reference track(s) from
<song>/<stem_level_directory_name>[/<separation_model>]/<input|reference>/<stem>.wav, clamp to [-1, 1] and stack to
[4, 2, L] float32 tensors.

When args.normalize_input is set (the reference CLI's default) the INPUT stems go through the FX normaliser
(Audio_Effects_Normalizer, mixing_manipulator/data_normalization.py) with the features file
args.precomputed_normalization_feature and the effect order args.normalization_order (reference :559-563, :586-587).
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
            from ..mixing_manipulator.data_normalization import Audio_Effects_Normalizer
            self.normalization_chain = Audio_Effects_Normalizer(precomputed_feature_path=args.precomputed_normalization_feature,
                                                                STEMS=args.instruments, EFFECTS=args.normalization_order)

    def __len__(self):
        return len(self.data_dir_paths)

    def _stem(self, idx, which, inst, normalize=False):
        path = os.path.join(self.data_dir_paths[idx], self.stem_level_directory_name, which, inst + ".wav")
        wav = load_wav_segment(path, axis=0, sample_rate=self.args.sample_rate)
        if normalize:           # only the input stems are normalised (:586-587)
            wav = self.normalization_chain.normalize_audio(wav.transpose(), src=inst).transpose()
        return torch.clamp(torch.from_numpy(wav).float(), min=-1, max=1)

    def __getitem__(self, idx):
        inputs = [self._stem(idx, self.input_name, i, normalize=self.args.normalize_input) for i in self.instruments]
        refs = [self._stem(idx, self.reference_name, i) for i in self.instruments]
        dir_name = os.path.dirname(self.data_dir_paths[idx])
        if self.interpolate:
            refs_b = [self._stem(idx, self.reference_name_B, i) for i in self.instruments]
            return torch.stack(inputs, 0), torch.stack(refs, 0), torch.stack(refs_b, 0), dir_name
        return torch.stack(inputs, 0), torch.stack(refs, 0), dir_name

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

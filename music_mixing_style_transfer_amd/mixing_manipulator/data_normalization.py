"""'Audio effects chain normalisation' of the input stems (reference mixing_manipulator/data_normalization.py:19-172,
Audio_Effects_Normalizer), applied by Song_Dataset_Inference when --normalize_input is set (the reference CLI's default).

Same class, constants, method names and feature-file format as the reference:
    Audio_Effects_Normalizer(precomputed_feature_path, STEMS, EFFECTS).normalize_audio(audio [L, 2], src) -> [L, 2]
The per-effect control flow (65536-sample zero padding, the -40 dB gate, per-channel loops, un-padding) is the reference's;
the sample-rate arithmetic of every effect runs on the MI355X (see utils_data_normalization.py, normalization_imager.py,
fx_utils.py).  `precomputed_feature_path` may also be the feature dictionary itself.
"""
import numpy as np
import scipy.signal

from . import fx_utils
from .normalization_imager import normalize_imager
from .utils_data_normalization import amp_to_db, get_comp_matching, get_eq_matching


class Audio_Effects_Normalizer:
    def __init__(self, precomputed_feature_path, STEMS=["drums", "bass", "other", "vocals"],
                 EFFECTS=["eq", "compression", "imager", "loudness"]):
        self.STEMS = STEMS          # stems to be normalised
        self.EFFECTS = EFFECTS      # effects to be normalised, order matters
        self.SR = 44100
        self.SUBTYPE = "PCM_16"
        self.FFT_SIZE = 2 ** 16
        self.HOP_LENGTH = self.FFT_SIZE // 4
        self.NTAPS = 1001
        self.LUFS = -30
        self.MIN_DB = -40           # minimum amplitude to apply the matching to
        self.COMP_USE_EXPANDER = False
        self.COMP_PEAK_NORM = -10.0
        self.COMP_TRUE_PEAK = False
        self.COMP_PERCENTILE = 75
        self.COMP_MIN_TH = -40
        self.COMP_MAX_RATIO = 20
        per_stem = {"vocals": (7.5, 400.0, 4, 128), "drums": (10.0, 180.0, 6, 128), "bass": (10.0, 500.0, 5, 16),
                    "other": (15.0, 666.0, 4, 128)}
        self.comp_settings = {key: {} for key in self.STEMS}
        for key in self.comp_settings:
            if key in per_stem:
                a, r, ratio, n_mels = per_stem[key]
                self.comp_settings[key] = {"attack": a, "release": r, "ratio": ratio, "n_mels": n_mels}
        if isinstance(precomputed_feature_path, dict):
            features_mean = {k: dict(v) for k, v in precomputed_feature_path.items()}
        else:
            features_mean = np.load(precomputed_feature_path, allow_pickle="TRUE")[()]
        self.features_mean = self.smooth_feature(features_mean)
        self.haas_chain = None      # extension: the chain normalize_imager applies to an almost-mono stem (None = the reference's random Haas)

    def normalize_audio(self, audio, src):
        """audio [L, 2] numpy -> numpy like the reference; a float32 DEVICE tensor in stays on the device through all the effects
        (padded, gated, matched and un-padded there; one tensor back) - the same kernels on the same values, without the host
        round trips of the array interface."""
        assert src in self.STEMS
        normalized_audio = audio
        for cur_effect in self.EFFECTS:
            normalized_audio = self.normalize_audio_per_effect(normalized_audio, src=src, effect=cur_effect)
        return normalized_audio

    def _per_effect_device(self, audio, src, effect):
        import torch
        audio = audio.to(torch.float32)
        track = torch.nn.functional.pad(audio, (0, 0, self.FFT_SIZE, self.FFT_SIZE))
        assert track.dim() == 2
        if track.shape[1] == 1:
            track = track.repeat(1, 2)
        out = track.clone()
        max_db = amp_to_db(float(out.abs().max()))
        if max_db > self.MIN_DB:
            if effect == "eq":
                for ch in range(track.shape[1]):
                    out[:, ch] = get_eq_matching(out[:, ch].contiguous(), self.features_mean[effect][src], sr=self.SR, n_fft=self.FFT_SIZE,
                                                 hop_length=self.HOP_LENGTH, min_db=self.MIN_DB, ntaps=self.NTAPS, lufs=self.LUFS)
            elif effect == "compression":
                assert len(self.features_mean[effect][src]) == 2
                for ch in range(track.shape[1]):
                    try:
                        s = self.comp_settings[src]
                        matched = get_comp_matching(out[:, ch].contiguous(), self.features_mean[effect][src][0],
                                                    self.features_mean[effect][src][1], s["ratio"], s["attack"], s["release"],
                                                    sr=self.SR, min_db=self.MIN_DB, min_th=self.COMP_MIN_TH,
                                                    comp_peak_norm=self.COMP_PEAK_NORM, max_ratio=self.COMP_MAX_RATIO,
                                                    n_mels=s["n_mels"], true_peak=self.COMP_TRUE_PEAK,
                                                    percentile=self.COMP_PERCENTILE, expander=self.COMP_USE_EXPANDER)
                        out[:, ch] = matched[:, 0]
                    except Exception:               # the reference swallows every failure of a channel and stops (:131-132)
                        break
            elif effect == "loudness":
                out = fx_utils.lufs_normalize(out, self.SR, self.features_mean[effect][src], log=False)
            elif effect == "imager":
                mono_threshold = 0.99 if src == "bass" else 0.975
                out = normalize_imager(out, target_side_mid_bal=self.features_mean[effect][src], mono_threshold=mono_threshold,
                                       sr=self.SR, haas=self.haas_chain)
        return out[self.FFT_SIZE:self.FFT_SIZE + audio.shape[0]]

    def normalize_audio_per_effect(self, audio, src, effect):
        if not isinstance(audio, np.ndarray):
            return self._per_effect_device(audio, src, effect)
        audio = audio.astype(dtype=np.float32)
        audio_track = np.pad(audio, ((self.FFT_SIZE, self.FFT_SIZE), (0, 0)), mode="constant")
        assert len(audio_track.shape) == 2          # always expects two dimensions
        if audio_track.shape[1] == 1:               # mono to stereo with repeated channels
            audio_track = np.repeat(audio_track, 2, axis=-1)
        output_audio = audio_track.copy()
        max_db = amp_to_db(np.max(np.abs(output_audio)))
        if max_db > self.MIN_DB:
            if effect == "eq":
                for ch in range(audio_track.shape[1]):
                    matched = get_eq_matching(output_audio[:, ch], self.features_mean[effect][src], sr=self.SR, n_fft=self.FFT_SIZE,
                                              hop_length=self.HOP_LENGTH, min_db=self.MIN_DB, ntaps=self.NTAPS, lufs=self.LUFS)
                    np.copyto(output_audio[:, ch], matched, casting="same_kind")
            elif effect == "compression":
                assert len(self.features_mean[effect][src]) == 2
                for ch in range(audio_track.shape[1]):
                    try:
                        s = self.comp_settings[src]
                        matched = get_comp_matching(output_audio[:, ch], self.features_mean[effect][src][0],
                                                    self.features_mean[effect][src][1], s["ratio"], s["attack"], s["release"],
                                                    sr=self.SR, min_db=self.MIN_DB, min_th=self.COMP_MIN_TH,
                                                    comp_peak_norm=self.COMP_PEAK_NORM, max_ratio=self.COMP_MAX_RATIO,
                                                    n_mels=s["n_mels"], true_peak=self.COMP_TRUE_PEAK,
                                                    percentile=self.COMP_PERCENTILE, expander=self.COMP_USE_EXPANDER)
                        np.copyto(output_audio[:, ch], matched[:, 0], casting="same_kind")
                    except Exception:               # the reference swallows every failure of a channel and stops (:131-132)
                        break
            elif effect == "loudness":
                output_audio = fx_utils.lufs_normalize(output_audio, self.SR, self.features_mean[effect][src], log=False)
            elif effect == "imager":
                mono_threshold = 0.99 if src == "bass" else 0.975          # threshold of applying the Haas effect
                matched = normalize_imager(output_audio, target_side_mid_bal=self.features_mean[effect][src],
                                           mono_threshold=mono_threshold, sr=self.SR, haas=self.haas_chain)
                np.copyto(output_audio, matched, casting="same_kind")
        return output_audio[self.FFT_SIZE:self.FFT_SIZE + audio.shape[0]]

    def smooth_feature(self, feature_dict_):
        for effect in self.EFFECTS:
            for key in self.STEMS:
                if effect == "eq":
                    f = 401 if key in ("other", "vocals") else 151
                    feature_dict_[effect][key] = scipy.signal.savgol_filter(feature_dict_[effect][key], f, 1, mode="mirror")
                elif effect == "panning":
                    feature_dict_[effect][key] = scipy.signal.savgol_filter(feature_dict_[effect][key], 501, 1, mode="mirror")
        return feature_dict_

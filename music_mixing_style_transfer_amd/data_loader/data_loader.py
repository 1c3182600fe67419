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

from .loader_utils import load_wav_device, load_wav_segment, read_wav_raw


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
        # device: set by the runner (Mixing_Style_Transfer_Inference) to the GPU the stems are converted on - the stems are then decoded,
        # normalised and clamped THERE and returned as device tensors (the file's PCM bytes are what crosses PCIe); None = host arrays
        # like the reference.  workers (args.workers, the reference's DataLoader(num_workers=...)): > 0 = a background thread READS the next
        # song's wav files into memory while the current song is converted (file I/O only: decoding, normalising and everything else that
        # launches kernels stays on the consumer's thread and stream - see __iter__).
        self.device = None
        self.workers = int(getattr(args, "workers", 0) or 0)
        self._preread = None
        # dist (set by the runner when it runs on several ranks): the stems that need host / normaliser work are prepared by ONE rank each
        # (stem j by rank j % world) and broadcast - the normaliser, the expensive part of a song's preparation, is sharded by stems
        self.dist = None

    def _prepared_by_owner(self, idx, which, inst, j):
        """Input stem j of a multi-rank run: decoded + normalised by rank j % world, received by the others (one broadcast of [2, L])."""
        from .loader_utils import load_wav_length
        dist = self.dist
        rank, world = dist.get_rank(), dist.get_world_size()
        owner = j % world
        path = os.path.join(self.data_dir_paths[idx], self.stem_level_directory_name, which, inst + ".wav")
        if rank == owner:
            t = self._stem(idx, which, inst, normalize=True).contiguous()
        else:
            dev = self.device if (self.device is not None and dist.get_backend() == "nccl") else "cpu"
            t = torch.empty(2, load_wav_length(path), dtype=torch.float32, device=dev)
        if dist.get_backend() != "nccl" and t.is_cuda:          # test hook (gloo with several ranks on one GPU): through the host
            h = t.cpu()
            dist.broadcast(h, src=owner)
            return h.to(t.device)
        dist.broadcast(t, src=owner)
        return t.to(self.device) if self.device is not None else t

    def __len__(self):
        return len(self.data_dir_paths)

    def _stem(self, idx, which, inst, normalize=False):
        path = os.path.join(self.data_dir_paths[idx], self.stem_level_directory_name, which, inst + ".wav")
        if self.device is not None:
            try:
                wav = load_wav_device(path, self.device, sample_rate=self.args.sample_rate, preread=self._preread)      # float32 [2, L] on the device
            except ValueError as e:
                if "stereo files only" not in str(e):
                    raise
                wav = None
            if wav is not None:
                if normalize:
                    wav = self.normalization_chain.normalize_audio(wav.t().contiguous(), src=inst).t().contiguous()
                return torch.clamp(wav.float(), min=-1, max=1)
        wav = load_wav_segment(path, axis=0, sample_rate=self.args.sample_rate, preread=self._preread)
        if normalize:           # only the input stems are normalised (:586-587)
            wav = self.normalization_chain.normalize_audio(wav.transpose(), src=inst).transpose()
        return torch.clamp(torch.from_numpy(wav).float(), min=-1, max=1)

    def __getitem__(self, idx):
        if self.dist is not None and self.args.normalize_input:
            inputs = [self._prepared_by_owner(idx, self.input_name, inst, j) for j, inst in enumerate(self.instruments)]
        else:
            inputs = [self._stem(idx, self.input_name, i, normalize=self.args.normalize_input) for i in self.instruments]
        refs = [self._stem(idx, self.reference_name, i) for i in self.instruments]
        dir_name = os.path.dirname(self.data_dir_paths[idx])
        if self.interpolate:
            refs_b = [self._stem(idx, self.reference_name_B, i) for i in self.instruments]
            return torch.stack(inputs, 0), torch.stack(refs, 0), torch.stack(refs_b, 0), dir_name
        return torch.stack(inputs, 0), torch.stack(refs, 0), dir_name

    def _song_paths(self, idx):
        kinds = [self.input_name, self.reference_name] + ([self.reference_name_B] if self.interpolate else [])
        return [os.path.join(self.data_dir_paths[idx], self.stem_level_directory_name, k, inst + ".wav") for k in kinds for inst in self.instruments]

    def _read_song(self, idx):
        out = {}
        for p in self._song_paths(idx):
            try:
                out[p] = read_wav_raw(p)
            except Exception:           # a missing / malformed file raises where the reference raises: in __getitem__, on the consumer's thread
                pass
        return out

    def __iter__(self):
        # Multi-rank runs prepare their songs INLINE whatever args.workers says (the loader's broadcasts must stay on the consumer's thread, in
        # its order: collectives of two threads on one process group pair up differently on different ranks).
        if self.workers <= 0 or len(self) < 2 or self.dist is not None:
            for i in range(len(self)):
                yield self[i]
            return
        # One song ahead, FILE I/O ONLY: a thread reads song i + 1's wav files into memory while song i is converted; decoding, the
        # normaliser and the stacking run here, on the consumer's thread and stream.  (Round 3 prepared the whole next song on a thread
        # with its own stream: its ~150 small kernels and ~70 host round trips per song queue behind the converter's 1.5 ms launches, the
        # prepared song arrived later than preparing it inline takes - measured 0.342 s per song against 0.324 s without the thread.)
        # The executor's context joins the thread when the generator is closed early (an exception in inference(), a break).
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=1, thread_name_prefix="mst-prefetch") as pool:
            nxt = pool.submit(self._read_song, 0)
            for i in range(len(self)):
                self._preread = nxt.result()
                nxt = pool.submit(self._read_song, i + 1) if i + 1 < len(self) else None
                try:
                    item = self[i]
                finally:
                    self._preread = None
                yield item

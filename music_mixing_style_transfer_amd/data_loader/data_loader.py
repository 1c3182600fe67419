"""Inference dataset (reference data_loader/data_loader.py:545-603, Song_Dataset_Inference): for every song
directory <target_dir>/<song>/ load the 4 stems of the input and of the reference track(s) from
<song>/<stem_level_directory_name>[/<separation_model>]/<input|reference>/<stem>.wav, clamp to [-1, 1] and stack to
[4, 2, L] float32 tensors.

When args.normalize_input is set (the reference CLI's default) the INPUT stems go through the FX normaliser
(Audio_Effects_Normalizer, mixing_manipulator/data_normalization.py) with the features file
args.precomputed_normalization_feature and the effect order args.normalization_order (reference :559-563, :586-587).
"""
import os
import queue
import threading
from glob import glob

import torch

from .loader_utils import load_wav_device, load_wav_segment


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
        # like the reference.  workers (args.workers, the reference's DataLoader(num_workers=...)): > 0 = songs are prepared by a
        # background thread one song ahead of the consumer.
        self.device = None
        self.workers = int(getattr(args, "workers", 0) or 0)
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
                wav = load_wav_device(path, self.device, sample_rate=self.args.sample_rate)            # float32 [2, L] on the device
            except ValueError as e:
                if "stereo files only" not in str(e):
                    raise
                wav = None
            if wav is not None:
                if normalize:
                    wav = self.normalization_chain.normalize_audio(wav.t().contiguous(), src=inst).t().contiguous()
                return torch.clamp(wav.float(), min=-1, max=1)
        wav = load_wav_segment(path, axis=0, sample_rate=self.args.sample_rate)
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

    def __iter__(self):
        # Multi-rank runs prepare their songs INLINE whatever args.workers says: __getitem__ issues collectives (the broadcast of the
        # normalised stems) and a prefetch thread would enqueue them on the default process group concurrently with the consumer's
        # all_gather / barrier of the previous song - in a different order on different ranks, which neither RCCL nor gloo allows.
        if self.workers <= 0 or len(self) < 2 or self.dist is not None:
            for i in range(len(self)):
                yield self[i]
            return
        # one song ahead: a thread decodes + normalises song i + 1 (its kernels run on its own stream) while song i is converted.  The
        # consumer may stop early (an exception in inference(), a break): the generator's finally sets `stop`, drains the queue so that a
        # blocked put returns, and joins the thread - no thread is left holding a song of device tensors.
        q = queue.Queue(maxsize=1)
        stop = threading.Event()

        def put(msg):
            while not stop.is_set():
                try:
                    q.put(msg, timeout=0.1)
                    return True
                except queue.Full:
                    pass
            return False

        def produce():
            try:
                if self.device is not None:
                    torch.cuda.set_device(self.device)
                    stream = torch.cuda.Stream(self.device)
                for i in range(len(self)):
                    if stop.is_set():
                        return
                    if self.device is not None:
                        with torch.cuda.stream(stream):
                            item = self[i]
                        stream.synchronize()
                    else:
                        item = self[i]
                    if not put(("item", item)):
                        return
                put(("done", None))
            except BaseException as e:          # surfaces in the consumer
                put(("error", e))
        worker = threading.Thread(target=produce, daemon=True, name="mst-prefetch")
        worker.start()
        try:
            while True:
                kind, payload = q.get()
                if kind == "done":
                    return
                if kind == "error":
                    raise payload
                if self.device is not None:         # made on the producer's stream, used on the consumer's: tell the caching allocator
                    for t in payload:
                        if isinstance(t, torch.Tensor) and t.is_cuda:
                            t.record_stream(torch.cuda.current_stream(self.device))
                yield payload
        finally:
            stop.set()
            try:
                while True:
                    q.get_nowait()
            except queue.Empty:
                pass
            worker.join(timeout=60)

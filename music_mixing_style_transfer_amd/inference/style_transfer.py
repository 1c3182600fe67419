"""Music mixing style transfer inference on MI355X - the orchestration of inference/style_transfer.py
(reference :27-177 Mixing_Style_Transfer_Inference, :181-270 interpolation, :344-389 arguments).

Same command-line flags and defaults, same directory layout, same checkpoint format, same configuration record; the
two networks run on libmst_hip.so.  Run plainly it drives one GPU; launched under
`python -m torch.distributed.run --nproc-per-node N` with an initialised process group (main() initialises "nccl" when
WORLD_SIZE > 1) every stem's segments are sharded across the N GPUs (`inference/engine.py::StyleTransferEngine.transfer_stem`:
one all-gather of segment embeddings, canonical-order mean, so the result does not depend on N) and every rank writes its own time
range of the output files.  The input FX normaliser (--normalize_input True, the reference's default) is implemented
(mixing_manipulator/data_normalization.py; its BS.1770 meter and onset detector are restatements: parity unpinned).
Not implemented: Demucs separation - the reference shells out to `demucs`; pass --do_not_separate True with the stems already
under <song>/separated/{input,reference}/ (the default --do_not_separate False raises NotImplementedError).

    python -m music_mixing_style_transfer_amd.inference.style_transfer --target_dir ./samples/style_transfer/ \
        --ckpt_path_enc FXencoder_ps.pt --ckpt_path_conv MixFXcloner_ps.pt --do_not_separate True
"""
import argparse
import os
from collections import OrderedDict

import numpy as np
import torch
import yaml

from ..data_loader import SlicedWavWriter, Song_Dataset_Inference, pcm16_device, save_wav_pcm16
from ..networks import FXencoder, TCNModel
from . import segmentation as seg
from .engine import embedding_mean

_HERE = os.path.dirname(os.path.abspath(__file__))


class Mixing_Style_Transfer_Inference:
    def __init__(self, args, trained_w_ddp=True):
        if args.inference_device != "cpu" and torch.cuda.is_available():
            self.device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) if self._world() is not None else 0)
            torch.cuda.set_device(self.device)
        else:
            raise RuntimeError("this build runs the networks on an MI355X only (no CPU path); no GPU is visible")
        if not args.do_not_separate:                 # checked before any model is built or checkpoint read
            raise NotImplementedError("source separation (demucs) is not part of this build: pass --do_not_separate True "
                                      "and provide the separated stems")
        self.args = args
        self.segment_length = args.segment_length
        self.batch_size = args.batch_size
        self.sample_rate = 44100
        self.output_dir = args.target_dir if args.output_dir is None else args.output_dir
        self.target_dir = args.target_dir
        precision = getattr(args, "precision", "fp32")

        self.models = {}
        self.models["effects_encoder"] = FXencoder(args.cfg_encoder).to(self.device)
        c = args.cfg_converter
        self.models["mixing_converter"] = TCNModel(
            nparams=c["condition_dimension"], ninputs=2, noutputs=2, nblocks=c["nblocks"],
            dilation_growth=c["dilation_growth"], kernel_size=c["kernel_size"], channel_width=c["channel_width"],
            stack_size=c["stack_size"], cond_dim=c["condition_dimension"], causal=c["causal"]).to(self.device)
        for m in self.models.values():
            m.precision = precision
            m.eval()
        self.reload_weights({"effects_encoder": args.ckpt_path_enc, "mixing_converter": args.ckpt_path_conv},
                            ddp=trained_w_ddp)
        self.data_loader = Song_Dataset_Inference(args)
        self.data_loader.device = self.device          # stems are decoded / normalised on the GPU and stay there
        self.data_loader.dist = self._world()          # several ranks: input stem j is normalised by rank j % N and broadcast
        if self._world() is None or self._world().get_rank() == 0:
            self.save_args(args)

    def reload_weights(self, ckpt_paths, ddp=True):
        for name, model in self.models.items():
            checkpoint = torch.load(ckpt_paths[name], map_location="cpu")
            state = OrderedDict()
            for k, v in checkpoint["model"].items():
                state[k[7:] if ddp else k] = v          # strip 'module.' of DDP-trained checkpoints
            model.load_state_dict(state)                 # strict, like the reference
            print(f"---reloaded checkpoint weights : {name} ---")

    def save_args(self, params):
        """The run's arguments, grouped like the command line's help, into
        <output_dir>style_transfer_inference_configurations.txt (reference style_transfer.py:305-322)."""
        info = "\n[args]\n"
        for group in build_parser()._action_groups:
            if group.title in ("positional arguments", "optional arguments", "options"):
                continue
            info += f"  {group.title} ({len(group._group_actions)})\n"
            for action in group._group_actions:
                info += f"      - {action.dest:20s}: {getattr(params, action.dest, None)}\n"
        info += "\n"
        os.makedirs(self.output_dir, exist_ok=True)
        with open(f"{self.output_dir}style_transfer_inference_configurations.txt", "w") as f:
            np.savetxt(f, [info], delimiter=" ", fmt="%s")

    # ---- hot loops -----------------------------------------------------------------------------
    @torch.no_grad()
    def _embed(self, batches):
        embs = [self.models["effects_encoder"](b.to(self.device)) for b in batches]
        return embedding_mean(seg.stack_embeddings(embs))

    @torch.no_grad()
    def _convert(self, batches, embedding_for_batch):
        outs = []
        for idx, b in enumerate(batches):
            emb = embedding_for_batch(idx)
            outs.append(self.models["mixing_converter"](b.to(self.device), emb.unsqueeze(0)))
        return outs

    # ---- multi-GPU: launched under torch.distributed.run, every rank owns a contiguous shard of each stem's segments ----
    @staticmethod
    def _world():
        import torch.distributed as dist
        return dist if (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1) else None

    def _engine(self):
        from .engine import StyleTransferEngine
        return StyleTransferEngine(self.models["effects_encoder"], self.models["mixing_converter"], device=self.device)

    def _host(self, stem):
        """Host stems go to the engine in page-locked memory (its passes overlap H2D / compute / D2H); device stems as they are."""
        if stem.device.type != "cpu":
            return stem
        return stem.pin_memory() if self.device.type == "cuda" and not stem.is_pinned() else stem

    def inference(self):
        """reference :112-177.  Every stem runs through StyleTransferEngine.transfer_stem: all of the stem's segments are
        in flight (passes bounded by the engine's sample budget, not by --batch_size, which only keeps the reference's
        ragged-`torch.stack` error), a rank reads / uploads / converts only its own shard of segments, and writes the time
        range it produced straight into the output files (SlicedWavWriter) - the audio is never gathered."""
        print("\n======= Start to inference music mixing style transfer =======")
        tag = "output" if self.args.normalize_input else "output_notnormed"
        a = self.args
        dist = self._world()
        rank = dist.get_rank() if dist is not None else 0
        eng = self._engine()
        for input_stems, reference_stems, dir_name in self.data_loader:
            print(f"---inference file name : {dir_name}---")
            out_dir = dir_name.replace(self.target_dir, self.output_dir)
            L = input_stems.shape[-1]
            names = [f"{inst}_{tag}.wav" for inst in a.instruments] if a.save_each_inst else []
            writers = {n: SlicedWavWriter(os.path.join(out_dir, n), L, 2, a.sample_rate) for n in names + [f"mixture_{tag}.wav"]}
            if dist is not None:
                if rank == 0:
                    os.makedirs(out_dir, exist_ok=True)
                    for w in writers.values():
                        w.create()
                dist.barrier()
            else:
                os.makedirs(out_dir, exist_ok=True)
            # all stems are ENQUEUED first, the files are written afterwards: converting a device-resident stem never waits for the host, so
            # the GPU works on stem i + 1 while stem i's PCM travels back and its file is written (written stem by stem the GPU idled
            # through five file writes per song)
            inst_outputs, t_range = [], (0, L)
            for i, inst in enumerate(a.instruments):
                print(f"\t{inst}...")
                seg.check_stackable(reference_stems[i].shape[-1], a.segment_length, a.segment_length_ref, a.batch_size)
                res = eng.transfer_stem(self._host(input_stems[i]), self._host(reference_stems[i]), a.segment_length,
                                        a.segment_length_ref, dir_name)
                stem_out, t_range = (res, (0, L)) if dist is None else res
                inst_outputs.append(stem_out)
            mixture = sum(inst_outputs)
            if a.save_each_inst:
                for inst, stem_out in zip(a.instruments, inst_outputs):
                    self._write(dist, writers[f"{inst}_{tag}.wav"], t_range[0], stem_out)
            self._write(dist, writers[f"mixture_{tag}.wav"], t_range[0], mixture)
            if dist is not None:
                dist.barrier()

    @staticmethod
    def _write(dist, writer, t0, data):
        """data [2, n] float tensor: the whole stem (one process: the plain wav writer) or this rank's time range starting at t0.
        A device tensor is rounded to 16-bit PCM on the device (round-half-even(x * 32767), clipped - the writer's own arithmetic) and
        only the int16 samples travel to the host."""
        if data.device.type != "cpu":
            pcm = pcm16_device(data.transpose(-1, -2).contiguous()).cpu().numpy()
        else:
            pcm = data.numpy().transpose(-1, -2)
        if dist is None:
            save_wav_pcm16(writer.path, pcm, writer.sr)
        else:
            writer.write(t0, pcm)

    def inference_interpolation(self):
        """reference :181-270.  Like inference(): every segment of a stem in flight through the engine, the segments sharded over the
        ranks (the two reference embeddings by one all-gather each), every rank writes the time range it produced.  The reference's
        bookkeeping is kept: the input is cut into `interpolate_segments` pieces of L // S + 1 samples, reference B is cut by
        segment_length (not _ref, :212), and the interpolation weight follows the BATCH index of a segment (:245)."""
        print("\n======= Start to inference interpolation examples =======")
        tag = "output_interpolation" if self.args.normalize_input else "output_notnormed_interpolation"
        a = self.args
        dist = self._world()
        rank = dist.get_rank() if dist is not None else 0
        eng = self._engine()
        S = a.interpolate_segments
        for input_stems, ref_a, ref_b, dir_name in self.data_loader:
            out_dir = dir_name.replace(self.target_dir, self.output_dir)
            L = input_stems.shape[-1]
            names = [f"{inst}_{tag}.wav" for inst in a.instruments] if a.save_each_inst else []
            writers = {n: SlicedWavWriter(os.path.join(out_dir, n), L, 2, a.sample_rate) for n in names + [f"mixture_{tag}.wav"]}
            if dist is not None:
                if rank == 0:
                    os.makedirs(out_dir, exist_ok=True)
                    for w in writers.values():
                        w.create()
                dist.barrier()
            else:
                os.makedirs(out_dir, exist_ok=True)
            inst_outputs, t_range = [], (0, L)
            for i, inst in enumerate(a.instruments):
                seg_len = input_stems[i].shape[1] // S + 1
                seg._check_duration(L, a.segment_length, dir_name)

                def emb_of(stem, cut):
                    if stem.shape[-1] > a.segment_length_ref:
                        seg._check_duration(stem.shape[-1], a.segment_length, dir_name)
                        n = seg.segment_count(stem.shape[-1], cut)
                        if n > a.batch_size and n % a.batch_size:        # the reference's torch.stack of ragged batches (:152)
                            raise RuntimeError(f"stack expects each tensor to be equal size, but got [{a.batch_size}, 2048] at entry 0 "
                                               f"and [{n % a.batch_size}, 2048] at entry {n // a.batch_size}")
                        return eng.mean_embedding(self._host(stem), n, cut)
                    return eng.mean_embedding(self._host(stem), 1, None)
                emb_a, emb_b = emb_of(ref_a[i], a.segment_length_ref), emb_of(ref_b[i], a.segment_length)

                def rows(p, k0, k1):
                    out = []
                    for k in range(k0, k1):
                        w = (S - 1 - k // a.batch_size) / (S - 1)          # weight by BATCH index, like the reference (:245)
                        out.append(w * emb_a + (1 - w) * emb_b)
                    return torch.stack(out)
                stem_out, t_range = eng.convert_segments(self._host(input_stems[i]), seg.segment_count(L, seg_len), seg_len, rows)
                inst_outputs.append(stem_out)
                if a.save_each_inst:
                    self._write(dist, writers[f"{inst}_{tag}.wav"], t_range[0], stem_out)
            self._write(dist, writers[f"mixture_{tag}.wav"], t_range[0], sum(inst_outputs))
            if dist is not None:
                dist.barrier()


def str2bool(v):
    if v.lower() in ("yes", "true", "t", "y", "1"):
        return True
    if v.lower() in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("Boolean value expected.")


def build_parser():
    root = os.path.dirname(os.path.dirname(_HERE))
    p = argparse.ArgumentParser()
    d = p.add_argument_group("Directory args")
    d.add_argument("--target_dir", type=str, default="./samples/style_transfer/")
    d.add_argument("--output_dir", type=str, default=None)
    d.add_argument("--input_file_name", type=str, default="input")
    d.add_argument("--reference_file_name", type=str, default="reference")
    d.add_argument("--reference_file_name_2interpolate", type=str, default="reference_B")
    d.add_argument("--ckpt_path_enc", type=str, default=os.path.join(root, "weights", "FXencoder_ps.pt"))
    d.add_argument("--ckpt_path_conv", type=str, default=os.path.join(root, "weights", "MixFXcloner_ps.pt"))
    d.add_argument("--precomputed_normalization_feature", type=str,
                   default=os.path.join(root, "weights", "musdb18_fxfeatures_eqcompimagegain.npy"))
    i = p.add_argument_group("Inference args")
    i.add_argument("--sample_rate", type=int, default=44100)
    i.add_argument("--segment_length", type=int, default=2 ** 19)
    i.add_argument("--segment_length_ref", type=int, default=2 ** 19)
    i.add_argument("--instruments", type=str2bool, default=["drums", "bass", "other", "vocals"])
    i.add_argument("--stem_level_directory_name", type=str, default="separated")
    i.add_argument("--save_each_inst", type=str2bool, default=False)
    i.add_argument("--do_not_separate", type=str2bool, default=False)
    i.add_argument("--separation_model", type=str, default="mdx_extra")
    i.add_argument("--normalize_input", type=str2bool, default=True)
    i.add_argument("--normalization_order", type=str2bool, default=["loudness", "eq", "compression", "imager", "loudness"])
    i.add_argument("--interpolation", type=str2bool, default=False)
    i.add_argument("--interpolate_segments", type=int, default=30)
    v = p.add_argument_group("Device args")
    v.add_argument("--workers", type=int, default=1)
    v.add_argument("--inference_device", type=str, default="gpu")
    v.add_argument("--batch_size", type=int, default=1)
    v.add_argument("--separation_device", type=str, default="cpu")
    v.add_argument("--precision", type=str, default="fp32", choices=["fp32", "bf16", "bf16x3"],
                   help="fp32 = exact-fp32 MFMA (parity with the reference), bf16 = throughput mode, bf16x3 = split-bf16 MFMA "
                        "(fp32-class accuracy, <= 1e-4 on the waveform, at about 3x the fp32 rate)")
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    with open(os.path.join(os.path.dirname(_HERE), "networks", "configs.yaml")) as f:
        configs = yaml.full_load(f)
    args.cfg_encoder = configs["Effects_Encoder"]["default"]
    args.cfg_converter = configs["TCN"]["default"]
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:             # one process per GPU (torch.distributed.run): RCCL over xGMI
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl")
    runner = Mixing_Style_Transfer_Inference(args)
    if args.interpolation:
        runner.inference_interpolation()
    else:
        runner.inference()


if __name__ == "__main__":
    main()

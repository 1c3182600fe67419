"""Music mixing style transfer inference on MI355X - the orchestration of inference/style_transfer.py
(reference :27-177 Mixing_Style_Transfer_Inference, :181-270 interpolation, :344-389 arguments).

Same command-line flags and defaults, same directory layout, same checkpoint format, same configuration record; the
two networks run on libmst_hip.so.  Run plainly it drives one GPU, batch by batch like the reference; launched under
`python -m torch.distributed.run --nproc-per-node N` with an initialised process group (main() initialises "nccl" when
WORLD_SIZE > 1) every stem's segments are sharded across the N GPUs (`inference/engine.py::StyleTransferEngine.transfer_stem`:
one all-gather of segment embeddings, canonical-order mean, so the result does not depend on N) and rank 0 writes the files.
Not implemented: Demucs separation (pass --do_not_separate True) and the input FX normaliser
(--normalize_input False); both are outside the accelerated hot path.

    python -m music_mixing_style_transfer_amd.inference.style_transfer --target_dir ./samples/style_transfer/ \
        --ckpt_path_enc FXencoder_ps.pt --ckpt_path_conv MixFXcloner_ps.pt --do_not_separate True --normalize_input False
"""
import argparse
import os
from collections import OrderedDict

import numpy as np
import torch
import yaml

from ..data_loader import Song_Dataset_Inference, save_wav_pcm16
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
        self.save_args(args)
        if not args.do_not_separate:
            raise NotImplementedError("source separation (demucs) is not part of this build: pass --do_not_separate True "
                                      "and provide the separated stems")

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

    @torch.no_grad()
    def _transfer_sharded(self, dist, input_stem, reference_stem, name):
        """One stem over all ranks (StyleTransferEngine.transfer_stem: sharded encoder pass, one all-gather of segment
        embeddings, sharded converter pass), then the converted segments are gathered so that every rank holds the stem."""
        from .engine import StyleTransferEngine
        a = self.args
        eng = StyleTransferEngine(self.models["effects_encoder"], self.models["mixing_converter"])
        out, (lo, hi) = eng.transfer_stem(input_stem.to(self.device), reference_stem.to(self.device), a.segment_length,
                                          a.segment_length_ref, name)
        world, n_seg = dist.get_world_size(), seg.segment_input(input_stem, name, a.segment_length, 1 << 30)[0].shape[0]
        counts = [seg.shard_range(n_seg, r, world) for r in range(world)]
        mx = max(h - l for l, h in counts)
        seg_len = out.shape[-1] if out.numel() else seg.segment_input(input_stem, name, a.segment_length, 1 << 30)[0].shape[-1]
        padded = torch.zeros(mx, 2, seg_len, dtype=torch.float32, device=self.device)
        padded[:hi - lo] = out
        gathered = torch.empty(world * mx, 2, seg_len, dtype=torch.float32, device=self.device)
        dist.all_gather_into_tensor(gathered, padded)
        full = torch.cat([gathered[r * mx:r * mx + (h - l)] for r, (l, h) in enumerate(counts)], 0)
        return seg.reassemble([full.cpu()], input_stem.shape[-1])

    def inference(self):
        print("\n======= Start to inference music mixing style transfer =======")
        tag = "output" if self.args.normalize_input else "output_notnormed"
        a = self.args
        dist = self._world()
        writer = dist is None or dist.get_rank() == 0
        for input_stems, reference_stems, dir_name in self.data_loader:
            print(f"---inference file name : {dir_name}---")
            out_dir = dir_name.replace(self.target_dir, self.output_dir)
            if writer:
                os.makedirs(out_dir, exist_ok=True)
            inst_outputs = []
            for i, inst in enumerate(a.instruments):
                print(f"\t{inst}...")
                if dist is not None:
                    stem_out = self._transfer_sharded(dist, input_stems[i], reference_stems[i], dir_name).numpy()
                else:
                    in_b = seg.segment_input(input_stems[i], dir_name, a.segment_length, a.batch_size)
                    ref_b = seg.segment_reference(reference_stems[i], dir_name, a.segment_length, a.segment_length_ref, a.batch_size)
                    emb = self._embed(ref_b)
                    outs = self._convert(in_b, lambda idx: emb)
                    stem_out = seg.reassemble([o.cpu() for o in outs], input_stems[i].shape[-1]).numpy()
                inst_outputs.append(stem_out)
                if a.save_each_inst and writer:
                    save_wav_pcm16(os.path.join(out_dir, f"{inst}_{tag}.wav"), stem_out.transpose(-1, -2), a.sample_rate)
            mix = sum(inst_outputs)
            if writer:
                save_wav_pcm16(os.path.join(out_dir, f"mixture_{tag}.wav"), mix.transpose(-1, -2), a.sample_rate)

    def inference_interpolation(self):
        print("\n======= Start to inference interpolation examples =======")
        tag = "output_interpolation" if self.args.normalize_input else "output_notnormed_interpolation"
        a = self.args
        dist = self._world()
        writer = dist is None or dist.get_rank() == 0        # interpolation is not sharded: every rank computes, rank 0 writes
        for input_stems, ref_a, ref_b, dir_name in self.data_loader:
            out_dir = dir_name.replace(self.target_dir, self.output_dir)
            if writer:
                os.makedirs(out_dir, exist_ok=True)
            inst_outputs = []
            for i, inst in enumerate(a.instruments):
                seg_len = input_stems[i].shape[1] // a.interpolate_segments + 1
                in_b = seg.batchwise_segmentization(input_stems[i], dir_name, seg_len, a.batch_size, min_length=a.segment_length)
                ra = seg.batchwise_segmentization(ref_a[i], dir_name, a.segment_length_ref, a.batch_size, min_length=a.segment_length) \
                    if ref_a[i].shape[-1] > a.segment_length_ref else [ref_a[i].unsqueeze(0)]
                # the reference cuts reference B by segment_length (not _ref) - kept (style_transfer.py:212)
                rb = seg.batchwise_segmentization(ref_b[i], dir_name, a.segment_length, a.batch_size, min_length=a.segment_length) \
                    if ref_b[i].shape[-1] > a.segment_length_ref else [ref_b[i].unsqueeze(0)]
                emb_a, emb_b = self._embed(ra), self._embed(rb)
                S = a.interpolate_segments

                def emb_for(idx):
                    w = (S - 1 - idx) / (S - 1)                  # weight by BATCH index, like the reference (:245)
                    return w * emb_a + (1 - w) * emb_b
                outs = self._convert(in_b, emb_for)
                stem_out = seg.reassemble([o.cpu() for o in outs], input_stems[i].shape[-1]).numpy()
                inst_outputs.append(stem_out)
                if a.save_each_inst and writer:
                    save_wav_pcm16(os.path.join(out_dir, f"{inst}_{tag}.wav"), stem_out.transpose(-1, -2), a.sample_rate)
            if writer:
                save_wav_pcm16(os.path.join(out_dir, f"mixture_{tag}.wav"), sum(inst_outputs).transpose(-1, -2), a.sample_rate)


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
    v.add_argument("--precision", type=str, default="fp32", choices=["fp32", "bf16"],
                   help="fp32 = exact-fp32 MFMA (parity with the reference), bf16 = throughput mode")
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

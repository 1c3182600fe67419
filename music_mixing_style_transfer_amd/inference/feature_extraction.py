"""FX-embedding extraction on MI355X - the reference's inference/feature_extraction.py (:20-193): for every *.wav under
--target_dir, cut the song into --segment_length pieces (last one zero padded; an exact multiple gets one extra
all-zero segment, like the reference), run the FXencoder on every segment, average the embeddings with torch.cat
semantics (ragged last batch allowed, :105) and save `<name>_fx_embedding.npy`.  Mono files are duplicated to stereo
(:82-83).  Same flags and defaults, except that the networks only run on the GPU (`--inference_device gpu`).

    python -m music_mixing_style_transfer_amd.inference.feature_extraction --target_dir ./samples/ --ckpt_path_enc FXencoder_ps.pt
"""
import argparse
import os
from collections import OrderedDict
from glob import glob

import numpy as np
import torch
import yaml

from ..data_loader import load_wav_segment
from ..networks import FXencoder
from . import segmentation as seg
from .engine import embedding_mean

_HERE = os.path.dirname(os.path.abspath(__file__))


class FXencoder_Inference:
    def __init__(self, args, trained_w_ddp=True):
        if not torch.cuda.is_available():
            raise RuntimeError("this build runs the FXencoder on an MI355X only (no CPU path); no GPU is visible")
        self.device = torch.device("cuda:0")
        self.segment_length = args.segment_length
        self.batch_size = args.batch_size
        self.sample_rate = 44100
        self.output_dir = args.target_dir if args.output_dir is None else args.output_dir
        self.target_dir = args.target_dir
        self.models = {"effects_encoder": FXencoder(args.cfg_encoder).to(self.device).eval()}
        self.models["effects_encoder"].precision = getattr(args, "precision", "fp32")
        self.reload_weights({"effects_encoder": args.ckpt_path_enc}, ddp=trained_w_ddp)
        self.save_args(args)

    def save_args(self, params):
        """<output_dir>feature_extraction_inference_configurations.txt, grouped like the help text (reference :144-161)."""
        info = "\n[args]\n"
        for group in build_parser()._action_groups:
            if group.title in ("positional arguments", "optional arguments", "options"):
                continue
            info += f"  {group.title} ({len(group._group_actions)})\n"
            for action in group._group_actions:
                info += f"      - {action.dest:20s}: {getattr(params, action.dest, None)}\n"
        info += "\n"
        os.makedirs(self.output_dir, exist_ok=True)
        with open(f"{self.output_dir}feature_extraction_inference_configurations.txt", "w") as f:
            np.savetxt(f, [info], delimiter=" ", fmt="%s")

    def reload_weights(self, ckpt_paths, ddp=True):
        for name, model in self.models.items():
            checkpoint = torch.load(ckpt_paths[name], map_location="cpu")
            model.load_state_dict(OrderedDict((k[7:] if ddp else k, v) for k, v in checkpoint["model"].items()))
            print(f"---reloaded checkpoint weights : {name} ---")

    @torch.no_grad()
    def embed_song(self, song, name="song"):
        """song: float [2, L] tensor -> averaged FX embedding, numpy [C]."""
        batches = seg.batchwise_segmentization(song, name, self.segment_length, self.batch_size)
        embs = [self.models["effects_encoder"](b.to(self.device)) for b in batches]
        return embedding_mean(torch.cat(embs, dim=0)).cpu().numpy()

    def save_averaged_embeddings(self):
        paths = glob(os.path.join(self.target_dir, "**", "*.wav"), recursive=True)
        for step, path in enumerate(paths):
            print(f"\nInference step : {step + 1}/{len(paths)}\n---current file path : {path}---")
            wav = load_wav_segment(path, axis=0)
            if wav.ndim == 1:
                wav = np.stack((wav, wav), axis=0)
            elif wav.shape[1] == 2:
                wav = wav.transpose()
            emb = self.embed_song(torch.from_numpy(wav).float(), path)
            out = path.replace(self.target_dir, self.output_dir).replace(".wav", "_fx_embedding.npy")
            os.makedirs(os.path.dirname(out), exist_ok=True)
            np.save(out, emb)


def build_parser():
    root = os.path.dirname(os.path.dirname(_HERE))
    p = argparse.ArgumentParser()
    d = p.add_argument_group("Directory args")
    d.add_argument("--target_dir", type=str, default="./samples/")
    d.add_argument("--output_dir", type=str, default=None)
    d.add_argument("--ckpt_path_enc", type=str, default=os.path.join(root, "weights", "FXencoder_ps.pt"))
    i = p.add_argument_group("Inference args")
    i.add_argument("--segment_length", type=int, default=44100 * 10)
    i.add_argument("--batch_size", type=int, default=1)
    i.add_argument("--inference_device", type=str, default="gpu")
    i.add_argument("--precision", type=str, default="fp32", choices=["fp32", "bf16"])
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    with open(os.path.join(os.path.dirname(_HERE), "networks", "configs.yaml")) as f:
        args.cfg_encoder = yaml.full_load(f)["Effects_Encoder"]["default"]
    FXencoder_Inference(args).save_averaged_embeddings()


if __name__ == "__main__":
    main()

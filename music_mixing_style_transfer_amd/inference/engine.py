"""The two hot loops of inference/style_transfer.py (:144-162) as one device-resident step, and their
multi-GPU form: segments are independent (eval-mode BatchNorm, per-segment zero padding), so each rank runs
the FXencoder / MixFXcloner on a contiguous shard of the segments and the only exchange is one all-gather
of the [n_seg_local, 2048] segment embeddings before the mean-pool (torch.distributed backend "nccl" = RCCL
over xGMI).  The mean is taken over ALL rows in canonical segment order, so it does not depend on the number
of GPUs.
"""
import ctypes as C

import torch

from .. import _lib
from ..networks import FXencoder, TCNModel
from . import segmentation as seg


def build_models(cfg_encoder, cfg_converter, device, precision="fp32"):
    """Construct the two networks exactly like Mixing_Style_Transfer_Inference.__init__ (style_transfer.py:47-57)."""
    enc = FXencoder(cfg_encoder).to(device)
    tcn = TCNModel(nparams=cfg_converter["condition_dimension"], ninputs=2, noutputs=2,
                   nblocks=cfg_converter["nblocks"], dilation_growth=cfg_converter["dilation_growth"],
                   kernel_size=cfg_converter["kernel_size"], channel_width=cfg_converter["channel_width"],
                   stack_size=cfg_converter["stack_size"], cond_dim=cfg_converter["condition_dimension"],
                   causal=cfg_converter["causal"]).to(device)
    tcn.precision = precision
    enc.precision = precision
    enc.eval()
    tcn.eval()
    return enc, tcn


def embedding_mean(emb):
    """[n, D] device tensor -> [D]: mean over rows in row order (mst_embedding_mean)."""
    b = _lib.lib()
    emb = emb.contiguous()
    out = torch.empty(emb.shape[1], dtype=torch.float32, device=emb.device)
    st = C.c_void_p(torch.cuda.current_stream(emb.device).cuda_stream) if emb.is_cuda else C.c_void_p(0)
    b.check(b.mst_embedding_mean(emb.data_ptr(), emb.shape[0], emb.shape[1], out.data_ptr(), st), "mst_embedding_mean")
    return out


class StyleTransferEngine:
    def __init__(self, encoder, converter, group=None):
        self.enc, self.tcn = encoder, converter
        self.group = group
        import torch.distributed as dist
        self.dist = dist if (dist.is_available() and dist.is_initialized()) else None
        self.world = self.dist.get_world_size(group) if self.dist else 1
        self.rank = self.dist.get_rank(group) if self.dist else 0

    @torch.no_grad()
    def reference_embedding(self, ref_segments_local, counts=None):
        """ref_segments_local [s_local, 2, L] (this rank's contiguous shard) -> (emb_local, mean over all ranks' rows).
        counts: rows per rank when shards are uneven (None: every rank holds s_local rows)."""
        emb = self.enc(ref_segments_local)
        if self.world == 1:
            return emb, embedding_mean(emb)
        if counts is None:
            allemb = torch.empty(self.world * emb.shape[0], emb.shape[1], dtype=emb.dtype, device=emb.device)
            self.dist.all_gather_into_tensor(allemb, emb.contiguous(), group=self.group)
        else:
            mx = max(counts)
            padded = torch.zeros(mx, emb.shape[1], dtype=emb.dtype, device=emb.device)
            padded[:emb.shape[0]] = emb
            gathered = torch.empty(self.world * mx, emb.shape[1], dtype=emb.dtype, device=emb.device)
            self.dist.all_gather_into_tensor(gathered, padded, group=self.group)
            allemb = torch.cat([gathered[r * mx:r * mx + counts[r]] for r in range(self.world)], 0)
        return emb, embedding_mean(allemb)

    @torch.no_grad()
    def step(self, ref_segments_local, in_segments_local, counts=None):
        """One pass of the hot path over this rank's shard: encoder -> (all-gather) -> mean -> converter."""
        _, emb_avg = self.reference_embedding(ref_segments_local, counts)
        return self.tcn(in_segments_local, emb_avg.unsqueeze(0)), emb_avg

    @torch.no_grad()
    def transfer_stem(self, input_stem, reference_stem, segment_length, segment_length_ref, song_name="song"):
        """One stem end to end on this rank's shard of segments (style_transfer.py:123-169, batch = all segments).
        input_stem / reference_stem: [2, L] device tensors (identical on every rank).  Returns the converted
        stem [2, L_in] on every rank when world == 1, else this rank's slice plus its (lo, hi) segment range."""
        in_b = seg.segment_input(input_stem, song_name, segment_length, 1 << 30)[0]
        ref_b = seg.segment_reference(reference_stem, song_name, segment_length, segment_length_ref, 1 << 30)[0]
        rlo, rhi = seg.shard_range(ref_b.shape[0], self.rank, self.world)
        ilo, ihi = seg.shard_range(in_b.shape[0], self.rank, self.world)
        counts = [seg.shard_range(ref_b.shape[0], r, self.world) for r in range(self.world)]
        counts = [hi - lo for lo, hi in counts]
        if rhi > rlo:
            emb = self.enc(ref_b[rlo:rhi].contiguous())
        else:
            emb = torch.zeros(0, self.tcn.hparams.cond_dim, dtype=torch.float32, device=input_stem.device)
        if self.world > 1:
            mx = max(counts)
            padded = torch.zeros(mx, emb.shape[1], dtype=emb.dtype, device=emb.device)
            padded[:emb.shape[0]] = emb
            gathered = torch.empty(self.world * mx, emb.shape[1], dtype=emb.dtype, device=emb.device)
            self.dist.all_gather_into_tensor(gathered, padded, group=self.group)
            emb = torch.cat([gathered[r * mx:r * mx + counts[r]] for r in range(self.world)], 0)
        emb_avg = embedding_mean(emb)
        out = self.tcn(in_b[ilo:ihi].contiguous(), emb_avg.unsqueeze(0)) if ihi > ilo else in_b[:0]
        if self.world == 1:
            return seg.reassemble([out], input_stem.shape[-1])
        return out, (ilo, ihi)

"""The two hot loops of inference/style_transfer.py (:144-162) as one device-resident step, and their
multi-GPU form: segments are independent (eval-mode BatchNorm, per-segment zero padding), so each rank runs
the FXencoder / MixFXcloner on a contiguous shard of the segments and the only exchange on the data path is one
all-gather of the [n_seg_local, 2048] segment embeddings before the mean-pool (torch.distributed backend "nccl" =
RCCL over xGMI).  The mean is taken over ALL rows in canonical segment order, so it does not depend on the number
of GPUs.

`transfer_stem` is the whole-stem pipeline (style_transfer.py:123-169 with every segment of the stem in flight):

  * a rank touches ONLY its own shard: segments [lo, hi) of the stem are cut from the host (or device) stem by
    integer bookkeeping - nothing is padded, copied or uploaded for segments other ranks own;
  * host stems (pinned memory) are uploaded on a copy stream in passes of at most `pass_samples` samples
    (default 2**23 = 64 segments of 131072) while the previous pass computes; results travel back the same way
    on a second copy stream, so H2D, compute and D2H overlap and the networks' workspace is bounded by one pass
    (2 * pass * 128 * 4 B = 8.6 GB fp32, 4.3 GB bf16) however long the stem is;
  * outputs stay sharded: a rank returns the time range [t_lo, t_hi) it produced (the style_transfer runner
    writes each rank's range straight into the output file).  `gather_stem` collects the ranges on ONE rank for
    callers that want the whole stem as a tensor - never an all-gather to everyone.
"""
import torch

from .. import _lib
from ..networks import FXencoder, TCNModel
from . import segmentation as seg

PASS_SAMPLES = 1 << 23          # samples per network pass (64 segments of 131072 / 16 segments of 2**19)


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
    with b.device_ctx(emb):
        out = torch.empty(emb.shape[1], dtype=torch.float32, device=emb.device)
        b.check(b.mst_embedding_mean(emb.data_ptr(), emb.shape[0], emb.shape[1], out.data_ptr(), b.stream_ptr(emb)),
                "mst_embedding_mean")
    return out


class _ShardFeed:
    """Segments [lo, hi) of a [2, L] stem as device batches [b, 2, seg_len] (zero padded past L), b <= per_pass.

    Device-resident stems are sliced in place.  Host stems are uploaded pass by pass on `copy_stream`: the rows of a
    pass are two contiguous host ranges (one per channel), so the copies are plain 1-d transfers straight out of the
    caller's (ideally pinned) memory; the [2, b*seg] -> [b, 2, seg] re-layout is a small device copy.  One pass is
    always in flight ahead of the consumer."""

    def __init__(self, stem, seg_len, lo, hi, per_pass, device, copy_stream):
        self.stem, self.seg, self.lo, self.hi = stem, seg_len, lo, hi
        self.per_pass, self.device, self.copy_stream = max(1, per_pass), device, copy_stream
        self.L = stem.shape[-1]

    def _ranges(self):
        return [(k, min(self.hi, k + self.per_pass)) for k in range(self.lo, self.hi, self.per_pass)]

    def _cut_resident(self, k0, k1):
        a, b = k0 * self.seg, min(k1 * self.seg, self.L)
        n = (k1 - k0) * self.seg
        slab = self.stem[:, a:b]
        if b - a < n:
            slab = torch.cat((slab, torch.zeros(2, n - (b - a), dtype=slab.dtype, device=slab.device)), dim=-1)
        return slab.reshape(2, k1 - k0, self.seg).transpose(0, 1).contiguous()

    def _upload(self, k0, k1):
        a, b = k0 * self.seg, min(k1 * self.seg, self.L)
        n = (k1 - k0) * self.seg
        cur = torch.cuda.current_stream(self.device)
        with torch.cuda.stream(self.copy_stream):
            slab = torch.empty(2, n, dtype=torch.float32, device=self.device)
            if b - a < n:
                slab[:, b - a:].zero_()
            for c in range(2):
                slab[c, :b - a].copy_(self.stem[c, a:b], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        slab.record_stream(cur)
        return slab, ev

    def __iter__(self):
        rng = self._ranges()
        if self.stem.device.type != "cpu" or self.copy_stream is None:
            for k0, k1 in rng:
                yield k0, k1, self._cut_resident(k0, k1)
            return
        nxt = self._upload(*rng[0]) if rng else None
        for i, (k0, k1) in enumerate(rng):
            slab, ev = nxt
            nxt = self._upload(*rng[i + 1]) if i + 1 < len(rng) else None       # keep one pass ahead
            torch.cuda.current_stream(self.device).wait_event(ev)
            yield k0, k1, slab.reshape(2, k1 - k0, self.seg).transpose(0, 1).contiguous()


class StyleTransferEngine:
    def __init__(self, encoder, converter, group=None, pass_samples=PASS_SAMPLES, device=None):
        self.enc, self.tcn = encoder, converter
        self.group = group
        self.device = device
        self.pass_samples = int(pass_samples)
        import torch.distributed as dist
        self.dist = dist if (dist.is_available() and dist.is_initialized()) else None
        self.world = self.dist.get_world_size(group) if self.dist else 1
        self.rank = self.dist.get_rank(group) if self.dist else 0
        self._streams = None

    # ---- device / stream plumbing ----------------------------------------------------------------
    def _device(self):
        return torch.device(self.device) if self.device is not None else next(self.tcn.parameters()).device

    def _copy_streams(self, device):
        if device.type != "cuda":
            return None, None
        if self._streams is None or self._streams[0].device != device:
            self._streams = (torch.cuda.Stream(device), torch.cuda.Stream(device))
        return self._streams

    # ---- the all-gather of segment embeddings -------------------------------------------------------
    def gather_embeddings(self, emb_local, counts):
        """[s_local, D] rows of this rank -> [sum(counts), D] rows of all ranks in segment order (one
        all_gather_into_tensor of equal-sized, zero-padded shards; the padding rows are dropped again)."""
        if self.world == 1:
            return emb_local
        mx = max(counts)
        padded = torch.zeros(mx, emb_local.shape[1], dtype=emb_local.dtype, device=emb_local.device)
        padded[:emb_local.shape[0]] = emb_local
        gathered = torch.empty(self.world * mx, emb_local.shape[1], dtype=emb_local.dtype, device=emb_local.device)
        self.dist.all_gather_into_tensor(gathered, padded, group=self.group)
        if all(c == mx for c in counts):
            return gathered
        return torch.cat([gathered[r * mx:r * mx + counts[r]] for r in range(self.world)], 0)

    @torch.no_grad()
    def reference_embedding(self, ref_segments_local, counts=None):
        """ref_segments_local [s_local, 2, L] (this rank's contiguous shard) -> (emb_local, mean over all ranks' rows).
        counts: rows per rank when shards are uneven (None: every rank holds s_local rows)."""
        emb = self.enc(ref_segments_local)
        if self.world == 1:
            return emb, embedding_mean(emb)
        counts = [emb.shape[0]] * self.world if counts is None else counts
        return emb, embedding_mean(self.gather_embeddings(emb, counts))

    @torch.no_grad()
    def step(self, ref_segments_local, in_segments_local, counts=None):
        """One pass of the hot path over this rank's shard: encoder -> (all-gather) -> mean -> converter."""
        _, emb_avg = self.reference_embedding(ref_segments_local, counts)
        return self.tcn(in_segments_local, emb_avg.unsqueeze(0)), emb_avg

    # ---- whole stems ----------------------------------------------------------------------------------
    def _per_pass(self, seg_len):
        return max(1, self.pass_samples // max(1, seg_len))

    @torch.no_grad()
    def stem_embedding(self, reference_stem, segment_length, segment_length_ref, song_name="song"):
        """Mean FX embedding [D] of a reference stem (style_transfer.py:133-153), sharded over the ranks."""
        n_ref, ref_len = seg.plan_reference(reference_stem.shape[-1], song_name, segment_length, segment_length_ref)
        return self.mean_embedding(reference_stem, n_ref, ref_len)

    @torch.no_grad()
    def mean_embedding(self, reference_stem, n_ref, ref_len):
        """Mean FX embedding [D] over the n_ref segments of ref_len samples of a stem (ref_len None: the stem as ONE item), the
        segments sharded over the ranks, one all-gather of the rows, canonical-order mean."""
        device = self._device()
        h2d, _ = self._copy_streams(device) if reference_stem.device.type == "cpu" else (None, None)
        if ref_len is None:                     # short reference: one [1, 2, L] item, encoded by the first rank
            counts = [1] + [0] * (self.world - 1)
            rows = [self.enc(reference_stem.unsqueeze(0).to(device))] if self.rank == 0 else []
        else:
            counts = [hi - lo for lo, hi in (seg.shard_range(n_ref, r, self.world) for r in range(self.world))]
            lo, hi = seg.shard_range(n_ref, self.rank, self.world)
            rows = [self.enc(x) for _, _, x in _ShardFeed(reference_stem, ref_len, lo, hi, self._per_pass(ref_len), device, h2d)]
        emb = torch.cat(rows, 0) if rows else torch.zeros(0, self.tcn.hparams.cond_dim, dtype=torch.float32, device=device)
        return embedding_mean(self.gather_embeddings(emb, counts))

    @torch.no_grad()
    def convert_stem(self, input_stem, embedding, segment_length, song_name="song", out=None):
        """This rank's part of the converted stem (style_transfer.py:126-132,157-169): segments [lo, hi) of the input
        stem through the MixFXcloner with `embedding` [D] (or a callable pass-index -> [D]).  Returns
        (y [2, t_hi - t_lo], (t_lo, t_hi)): the converted samples of the time range this rank owns, cropped to the
        stem's length; on the device for a device stem, in (pinned) host memory for a host stem (or `out`)."""
        n_in, in_len = seg.plan_input(input_stem.shape[-1], song_name, segment_length)
        cond = embedding if callable(embedding) else (lambda k: embedding)
        return self.convert_segments(input_stem, n_in, in_len, lambda p, k0, k1: cond(p).unsqueeze(0), out=out)

    @torch.no_grad()
    def convert_segments(self, input_stem, n_in, in_len, cond_rows, out=None):
        """This rank's shard of the n_in segments of in_len samples (in_len None: the stem as ONE item) through the MixFXcloner.
        cond_rows(pass index, k0, k1) -> the FiLM condition of segments [k0, k1): [1, D] (shared) or [k1 - k0, D] (one row per segment)."""
        device = self._device()
        L = input_stem.shape[-1]
        host = input_stem.device.type == "cpu" and device.type == "cuda"
        h2d, d2h = self._copy_streams(device) if host else (None, None)
        if in_len is None:                      # short input: one unsegmented item, converted by the first rank
            if self.rank != 0:
                return input_stem[:, :0], (0, 0)
            y = self.tcn(input_stem.unsqueeze(0).to(device), cond_rows(0, 0, 1))[0]
            return (y.to(input_stem.device) if host else y), (0, L)
        lo, hi = seg.shard_range(n_in, self.rank, self.world)
        t_lo, t_hi = min(L, lo * in_len), min(L, hi * in_len)
        if out is None:
            out = torch.empty(2, t_hi - t_lo, dtype=torch.float32, device=input_stem.device, pin_memory=host)
        pending = []
        for p, (k0, k1, x) in enumerate(_ShardFeed(input_stem, in_len, lo, hi, self._per_pass(in_len), device, h2d)):
            y = self.tcn(x, cond_rows(p, k0, k1))
            a, b = k0 * in_len - t_lo, min(k1 * in_len, L) - t_lo
            slab = y.transpose(0, 1).reshape(2, (k1 - k0) * in_len)            # cat(unbind(batch), time) (:165-166)
            if host:
                done = torch.cuda.Event()
                done.record(torch.cuda.current_stream(device))
                with torch.cuda.stream(d2h):
                    d2h.wait_event(done)
                    for c in range(2):
                        out[c, a:b].copy_(slab[c, :b - a], non_blocking=True)
                slab.record_stream(d2h)
                pending.append(slab)
            else:
                out[:, a:b] = slab[:, :b - a]
        if host:
            d2h.synchronize()
        return out, (t_lo, t_hi)

    @torch.no_grad()
    def transfer_stem(self, input_stem, reference_stem, segment_length, segment_length_ref, song_name="song"):
        """One stem end to end (style_transfer.py:123-169).  input_stem / reference_stem: [2, L] tensors, on the device
        or in host memory (pinned host memory overlaps its transfers with the networks); every rank passes the same
        stems and reads only its shard of them.  world == 1: returns the converted stem [2, L_in]; otherwise this
        rank's (y [2, t_hi - t_lo], (t_lo, t_hi)) - see `gather_stem` / the runner's sliced file writes."""
        emb_avg = self.stem_embedding(reference_stem, segment_length, segment_length_ref, song_name)
        y, rng = self.convert_stem(input_stem, emb_avg, segment_length, song_name)
        return y if self.world == 1 else (y, rng)

    def gather_stem(self, y_local, t_range, length, dst=0):
        """Collect the ranks' time ranges on rank `dst` only: returns [2, length] there, None elsewhere."""
        if self.world == 1:
            return y_local
        ranges = [None] * self.world
        self.dist.all_gather_object(ranges, tuple(t_range), group=self.group)
        mx = max(b - a for a, b in ranges)
        dev = self._device() if self.dist.get_backend(self.group) == "nccl" else torch.device("cpu")
        padded = torch.zeros(2, mx, dtype=torch.float32, device=dev)
        padded[:, :y_local.shape[-1]] = y_local
        parts = [torch.empty_like(padded) for _ in range(self.world)] if self.rank == dst else None
        self.dist.gather(padded, parts, dst=dst, group=self.group)
        if self.rank != dst:
            return None
        full = torch.empty(2, length, dtype=torch.float32, device=dev)
        for (a, b), p in zip(ranges, parts):
            full[:, a:b] = p[:, :b - a]
        return full

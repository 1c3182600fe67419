"""ORACLE (test infrastructure, NOT the product): segment bookkeeping of the reference, in numpy.

Restates inference/style_transfer.py:274-301 (batchwise_segmentization), :126-139 (segment /
no-segment thresholds), :152-153 (embedding mean-pool incl. padded segments), :165-169
(unbind/cat/crop reassembly).  Integer / index arithmetic: parity bar is bit-exact.
Pinned against the imported reference by tests/golden/make_golden.py (bookkeeping tables).
"""
import numpy as np


def segment_plan(length, segment_length, batch_size, min_length=None):
    """Index math of batchwise_segmentization(discard_last=False).

    Returns dict(pad, n_seg, batch_sizes).  Quirks preserved:
      * the duration assert compares with args.segment_length (min_length), not the
        segment_length parameter (style_transfer.py:275);
      * pad = seg - L % seg, i.e. a FULL extra all-zero segment when L is an exact multiple (:286).
    """
    if min_length is None:
        min_length = segment_length
    if length < min_length:
        raise AssertionError("Error : Insufficient duration!")
    pad = segment_length - length % segment_length
    n_seg = (length + pad) // segment_length
    sizes = [batch_size] * (n_seg // batch_size)
    if n_seg % batch_size:
        sizes.append(n_seg % batch_size)
    return {"pad": pad, "n_seg": n_seg, "batch_sizes": sizes}


def batchwise_segmentization(song, segment_length, batch_size, min_length=None):
    """song float [2, L] -> list of [b, 2, segment_length] arrays."""
    plan = segment_plan(song.shape[-1], segment_length, batch_size, min_length)
    padded = np.concatenate([song, np.zeros((song.shape[0], plan["pad"]), song.dtype)], axis=-1)
    out, cur = [], []
    for i in range(plan["n_seg"]):
        cur.append(padded[..., i * segment_length:(i + 1) * segment_length])
        if len(cur) == batch_size:
            out.append(np.stack(cur, 0))
            cur = []
    if cur:
        out.append(np.stack(cur, 0))
    return out


def input_batches(stem, segment_length, batch_size):
    """style_transfer.py:126-132: segment iff L > segment_length, else one [1,2,L] batch."""
    if stem.shape[-1] > segment_length:
        return batchwise_segmentization(stem, segment_length, batch_size, segment_length)
    return [stem[None]]


def reference_batches(stem, segment_length, segment_length_ref, batch_size):
    """style_transfer.py:133-139: segment iff L > 2*segment_length (cut by segment_length_ref)."""
    if stem.shape[-1] > 2 * segment_length:
        return batchwise_segmentization(stem, segment_length_ref, batch_size, segment_length)
    return [stem[None]]


def reassemble(batches, length):
    """style_transfer.py:165-169: per batch cat(unbind(dim0), dim=-1), cat batches on time, crop."""
    seq = [np.concatenate(list(b), axis=-1) for b in batches]
    return np.concatenate(seq, axis=-1)[:, :length]


def mean_embedding(emb_batches):
    """style_transfer.py:152-153: torch.stack (requires equal batch shapes) -> reshape -> mean(0)."""
    shapes = {e.shape for e in emb_batches}
    if len(shapes) != 1:
        raise RuntimeError("stack expects each tensor to be equal size")
    allemb = np.stack(emb_batches).reshape(-1, emb_batches[0].shape[-1])
    return allemb.astype(np.float32).mean(axis=0, dtype=np.float32)

"""Segment bookkeeping of the style-transfer inference loop (integer / index arithmetic, bit-exact with the
reference): inference/style_transfer.py:274-301 (batchwise_segmentization), :126-139 (when a stem is segmented),
:165-169 (reassembly), plus the contiguous rank sharding used for multi-GPU runs.
"""
import torch


def _check_duration(length, min_length, song_name):
    assert length >= min_length, \
        f"Error : Insufficient duration!\n\t \
                Target song's length is shorter than segment length.\n\t \
                Song name : {song_name}\n\t \
                Consider changing the 'segment_length' or song with sufficient duration"


def segment_count(length, segment_length):
    """Number of segments batchwise_segmentization cuts (discard_last=False): pad = seg - L % seg zeros are appended,
    so an exact multiple gets one extra all-zero segment (:286-293)."""
    return (length + segment_length - length % segment_length) // segment_length


def plan_input(length, song_name, segment_length):
    """(n_segments, segment_length) of an input stem, or (1, None) when it runs as one [1, 2, L] item (:126-132)."""
    if length > segment_length:
        _check_duration(length, segment_length, song_name)
        return segment_count(length, segment_length), segment_length
    return 1, None


def plan_reference(length, song_name, segment_length, segment_length_ref):
    """(n_segments, segment_length_ref) of a reference stem, or (1, None) when it is encoded whole (:133-139)."""
    if length > 2 * segment_length:
        _check_duration(length, segment_length, song_name)
        return segment_count(length, segment_length_ref), segment_length_ref
    return 1, None


def batchwise_segmentization(target_song, song_name, segment_length, batch_size, min_length=None, discard_last=False):
    """[2, L] tensor -> list of [b, 2, segment_length] batches (last one may be ragged).

    Reference behaviour kept on purpose:
      * the duration check compares with `min_length` = args.segment_length, not the segment_length argument;
      * pad = seg - L % seg: an exact multiple gets one extra all-zero segment.
    """
    min_length = segment_length if min_length is None else min_length
    _check_duration(target_song.shape[-1], min_length, song_name)
    if discard_last:
        target_song = target_song[:, :target_song.shape[-1] - target_song.shape[-1] % segment_length]
    else:
        pad_length = segment_length - target_song.shape[-1] % segment_length
        target_song = torch.cat((target_song, torch.zeros(2, pad_length, dtype=target_song.dtype,
                                                          device=target_song.device)), dim=-1)
    n_seg = target_song.shape[-1] // segment_length
    segs = target_song.reshape(target_song.shape[0], n_seg, segment_length).transpose(0, 1)   # [n_seg, 2, seg] view
    return [segs[i:i + batch_size].contiguous() for i in range(0, n_seg, batch_size)]


def segment_input(stem, song_name, segment_length, batch_size):
    """Input stems are segmented iff longer than segment_length, else run as one [1, 2, L] item (:126-132)."""
    if stem.shape[-1] > segment_length:
        return batchwise_segmentization(stem, song_name, segment_length, batch_size, min_length=segment_length)
    return [stem.unsqueeze(0)]


def segment_reference(stem, song_name, segment_length, segment_length_ref, batch_size):
    """Reference stems are segmented iff longer than 2 * segment_length, cut by segment_length_ref (:133-139)."""
    if stem.shape[-1] > 2 * segment_length:
        return batchwise_segmentization(stem, song_name, segment_length_ref, batch_size, min_length=segment_length)
    return [stem.unsqueeze(0)]


def reassemble(batches, length):
    """cat(unbind(batch), time) per batch, cat batches on time, crop to the stem's length (:165-169)."""
    seq = [torch.cat(torch.unbind(b, dim=0), dim=-1) for b in batches]
    return torch.cat(seq, dim=-1)[:, :length]


def stack_embeddings(emb_batches):
    """[n_batches] x [b, D] -> [n, D]; like the reference's torch.stack this requires equal batch shapes (:152)."""
    stacked = torch.stack(emb_batches)
    return stacked.reshape(stacked.shape[0] * stacked.shape[1], stacked.shape[2])


def check_stackable(ref_length, segment_length, segment_length_ref, batch_size):
    """The reference stacks the per-batch embeddings with torch.stack (:152), which raises when the last batch of
    reference segments is ragged; the runner keeps that error (feature_extraction.py uses torch.cat and does not)."""
    if ref_length <= 2 * segment_length:
        return
    n = segment_count(ref_length, segment_length_ref)
    if n > batch_size and n % batch_size:
        raise RuntimeError(f"stack expects each tensor to be equal size, but got [{batch_size}, 2048] at entry 0 and "
                           f"[{n % batch_size}, 2048] at entry {n // batch_size}")


def shard_range(n_items, rank, world_size):
    """Contiguous range [lo, hi) of rank `rank`: floor(r*S/N) .. floor((r+1)*S/N)."""
    return (rank * n_items) // world_size, ((rank + 1) * n_items) // world_size

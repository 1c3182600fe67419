"""FX-manipulator chains: the reference's `create_effects_augmentation_chain` and
`create_inst_effects_augmentation_chain` (mixing_manipulator/audio_effects_chain.py:16-164) over the device processors of
`common_audioeffects`.  Same effect names, probabilities, RMS-normalise rule (every effect except Gain and nested chains),
shuffle / parallel semantics and the drums low/high split of the reverb branch.

`AlgorithmicReverb` (requested by name, with `algorithmic=True`, or as the fallback when no impulse-response directory is given,
like the reference :44-49) runs on the device too; its comb / all-pass arithmetic is pymixconsole's, restated from the published
structure (parity unpinned).  `Expander`: the reference names it but defines no such class (NameError there); it raises here."""
import glob
import os

import numpy as np

from ..data_loader.loader_utils import load_wav_segment
from .common_audioeffects import (AlgorithmicReverb, AugmentationChain, Compressor, ConvolutionalReverb, Equaliser, Gain, MidSideImager, Panner,
                                  Parameter, ParameterList, Processor)


def load_impulse_responses(ir_dir_path, sample_rate=44100):
    """Impulse responses grouped by RT60 like the reference (:64-86): files live in
    `<ir_dir_path>*/RT60_avg/<rt60 range, e.g. 300-600>/<name>/impulse_response.wav`; one group per RT60 directory below
    3000 ms, all longer ones merged into a last group.  Entries are dicts with an 'impulse_response' loader returning
    [n_samples, n_channels] - the structure ConvolutionalReverb consumes."""
    groups, long_group = {}, []
    for rt_dir in sorted(glob.glob(f"{ir_dir_path}*/RT60_avg/[!0-]*")):
        rt = os.path.basename(rt_dir)
        entries = []
        for d in sorted(os.listdir(rt_dir)):
            f = os.path.join(rt_dir, d, "impulse_response.wav")
            if os.path.isfile(f):
                data = np.atleast_2d(load_wav_segment(f, axis=0, sample_rate=sample_rate)).T.copy()    # [n_samples, n_channels]
                entries.append({"impulse_response": (lambda data=data: data)})
        if not entries:
            continue
        if int(rt.split("-")[0]) < 3000:
            groups.setdefault(rt, []).extend(entries)
        else:
            long_group.extend(entries)
    return list(groups.values()) + [long_group]


def _make_processor(name, ir_dir_path, sample_rate):
    key = name.lower()
    if key == "gain":
        return Gain()
    if "eq" in key:
        return Equaliser(n_channels=2, sample_rate=sample_rate)
    if "comp" in key:
        return Compressor(sample_rate=sample_rate)
    if "expand" in key:
        raise NotImplementedError("'expand': the reference has no Expander class either (NameError there)")
    if "pan" in key:
        return Panner()
    if "image" in key:
        return MidSideImager()
    if "algorithmic" in key or ("reverb" in key and ir_dir_path is None):
        return AlgorithmicReverb(sample_rate=sample_rate)
    if "reverb" in key:
        return ConvolutionalReverb(load_impulse_responses(ir_dir_path, sample_rate), sample_rate)
    raise ValueError(f"make sure the target effects are in the Augment FX chain : received fx called {name}")


def create_effects_augmentation_chain(effects, ir_dir_path=None, sample_rate=44100, shuffle=False, parallel=False,
                                      parallel_weight_factor=None):
    """effects: names, (name, probability) tuples, Processor or AugmentationChain instances (probability 1 when omitted)."""
    entries = []
    for item in effects:
        fx, prob = (item[0], item[1]) if isinstance(item, tuple) else (item, 1)
        if not isinstance(fx, (AugmentationChain, Processor)):
            fx = _make_processor(fx, ir_dir_path, sample_rate)
        rms_normalize = not (isinstance(fx, AugmentationChain) or fx.name == "Gain")
        entries.append((fx, prob, rms_normalize))
    return AugmentationChain(fxs=entries, shuffle=shuffle, parallel=parallel, parallel_weight_factor=parallel_weight_factor)


def _shelf_cut(band, sample_rate):
    """A one-band Equaliser pinned at -50 dB / 100 Hz: 'high_shelf' keeps the lows, 'low_shelf' keeps the highs (:118-143)."""
    params = ParameterList()
    params.add(Parameter(band + "_gain", -50.0, "float", minimum=-50.0, maximum=-50.0))
    params.add(Parameter(band + "_freq", 100.0, "float", minimum=100.0, maximum=100.0))
    return Equaliser(n_channels=2, sample_rate=sample_rate, bands=[band], parameters=params)


def create_inst_effects_augmentation_chain(inst, apply_prob_dict, ir_dir_path=None, algorithmic=False, sample_rate=44100):
    """The FXmanipulator of one instrument: shuffled (eq, comp) -> shuffled (pan, imager) -> parallel reverb -> gain.
    Drums get the reverb on the band above 100 Hz (and, with 1 % of the probability, below it)."""
    reverb = "algorithmic" if algorithmic else "reverb"
    kw = dict(ir_dir_path=ir_dir_path, sample_rate=sample_rate)
    eq_comp = create_effects_augmentation_chain([("eq", apply_prob_dict["eq"]), ("comp", apply_prob_dict["comp"])], shuffle=True, **kw)
    pan_image = create_effects_augmentation_chain([("pan", apply_prob_dict["pan"]), ("imager", apply_prob_dict["imager"])],
                                                  shuffle=True, **kw)
    if inst == "drums":
        low = create_effects_augmentation_chain([_shelf_cut("high_shelf", sample_rate), (reverb, apply_prob_dict["reverb"] * 0.01)],
                                                parallel=True, parallel_weight_factor=0.8, **kw)
        high = create_effects_augmentation_chain([_shelf_cut("low_shelf", sample_rate), (reverb, apply_prob_dict["reverb"])],
                                                 parallel=True, parallel_weight_factor=0.6, **kw)
        reverb_branch = create_effects_augmentation_chain([low, high], **kw)
    else:
        reverb_branch = create_effects_augmentation_chain([(reverb, apply_prob_dict["reverb"])], parallel=True, **kw)
    return create_effects_augmentation_chain([eq_comp, pan_image, reverb_branch, ("gain", apply_prob_dict["gain"])], **kw)

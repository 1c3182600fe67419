"""Reproduces round 6's cross-stream finding (EXPERIMENTS.md E.4).  Two host threads, two handles, two streams: a DISTURBER loops one of our
bf16 networks on one stream while a VICTIM repeats one call on another; the victim's result is compared, bit for bit, with what it produces alone.

With packed-FP32 VALU instructions in the device code (`make -C music_mixing_style_transfer_amd/csrc clean all NOPK=`) the exact-fp32 TCN calls
return sporadically wrong values - low halves of v_pk_fma_f32 results in lanes 48-63, ~0.3 % off - whenever the disturber's kernels share their
CUs; alone, or beside torch's own kernels, they never do.  The shipped build (NOPK = -target-feature -packed-fp32-ops) prints zeros everywhere.

    python tools/repro_cross_stream.py"""
import os, sys, threading
import torch, yaml
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from music_mixing_style_transfer_amd import _lib
if "--lib" in sys.argv:          # another build of the library (e.g. one made with NOPK= : packed FP32 instructions on)
    _lib.set_default_binding(_lib.bind(sys.argv[sys.argv.index("--lib") + 1]))
    print("library:", sys.argv[sys.argv.index("--lib") + 1], flush=True)
from music_mixing_style_transfer_amd.inference import build_models
from music_mixing_style_transfer_amd.networks import FiLM
from music_mixing_style_transfer_amd.utils import synth
dev = torch.device("cuda:0")
with open(os.path.join(REPO, "music_mixing_style_transfer_amd", "networks", "configs.yaml")) as f:
    c = yaml.full_load(f)
enc_cfg, tcn_cfg = c["Effects_Encoder"]["default"], c["TCN"]["default"]
enc_sd, tcn_sd = synth.fxencoder_state_dict(enc_cfg, seed=0), synth.tcn_state_dict(seed=0)
def mk(prec):
    enc, tcn = build_models({k: (list(v) if isinstance(v, list) else v) for k, v in enc_cfg.items()}, tcn_cfg, dev, prec)
    enc.load_state_dict(enc_sd); tcn.load_state_dict(tcn_sd)
    return enc, tcn
A, Bm, Cm = mk("bf16"), mk("fp32"), mk("bf16")
xa = synth.synth_audio((3, 2, 40000), seed=300).to(dev)
xb = synth.synth_audio((3, 2, 41000), seed=301).to(dev)
emb = synth.synth_audio((1, 2048), seed=9).to(dev)
film = FiLM(2048, 128).to(dev)
xf = synth.synth_audio((3, 128, 41000), seed=5).to(dev)
A[1](xa, emb); lib = _lib.lib()
def a_tcn_onetile():
    return A[1](xa, emb)
victims = {"FiLM module (GEMV + elementwise)": lambda: film(xf, emb), "bf16 tcn probe, 1 block": lambda: Cm[1].forward_blocks(xb, emb, 1),
           "fp32 tcn probe, 1 block": lambda: Bm[1].forward_blocks(xb, emb, 1), "fp32 tcn forward": lambda: Bm[1](xb, emb), "bf16 tcn forward": lambda: Cm[1](xb, emb)}
disturbers = {"bf16 encoder": lambda: A[0](xa), "bf16 tcn (default forms)": a_tcn_onetile}
for dname, dfn in disturbers.items():
    for vname, v in victims.items():
        ref = v().clone(); dfn(); torch.cuda.synchronize()
        stop, res = [False], []
        def ta():
            with torch.cuda.stream(torch.cuda.Stream(device=dev)):
                while not stop[0]:
                    for _ in range(10):
                        dfn()
                    torch.cuda.current_stream().synchronize()
        def tb():
            with torch.cuda.stream(torch.cuda.Stream(device=dev)):
                for rep in range(8):
                    y = v()
                    torch.cuda.current_stream().synchronize()
                    res.append(int((y != ref).sum()))
                stop[0] = True
        ths = [threading.Thread(target=ta), threading.Thread(target=tb)]
        [t.start() for t in ths]; [t.join() for t in ths]
        print(f"disturber: {dname:26s} victim: {vname:34s} wrong elements per call: {res}", flush=True)

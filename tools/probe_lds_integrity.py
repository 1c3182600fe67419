"""Round 6, EXPERIMENTS.md E.4: LDS / register integrity of a probe kernel (tools/micro/lds_probe.hip: every workgroup keeps a pattern in 8 / 16 /
40 KB of LDS and three registers, rewrites it with 128-bit stores and re-reads it with 32 / 64 / 128-bit loads) while our kernels run on another
stream - nothing ever changes: the cross-stream fault is not a neighbour writing into this workgroup's LDS or registers.
    hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o tools/_ab/lds_probe.so tools/micro/lds_probe.hip && python tools/probe_lds_integrity.py"""
import ctypes as C, os, sys, threading
import torch, yaml
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from music_mixing_style_transfer_amd import _lib
from music_mixing_style_transfer_amd.inference import build_models
from music_mixing_style_transfer_amd.utils import synth
probe = C.CDLL(os.path.join(REPO, "tools", "_ab", "lds_probe.so"))
dev = torch.device("cuda:0")
with open(os.path.join(REPO, "music_mixing_style_transfer_amd", "networks", "configs.yaml")) as f:
    c = yaml.full_load(f)
enc_cfg, tcn_cfg = c["Effects_Encoder"]["default"], c["TCN"]["default"]
enc_sd, tcn_sd = synth.fxencoder_state_dict(enc_cfg, seed=0), synth.tcn_state_dict(seed=0)
def mk(prec):
    enc, tcn = build_models({k: (list(v) if isinstance(v, list) else v) for k, v in enc_cfg.items()}, tcn_cfg, dev, prec)
    enc.load_state_dict(enc_sd); tcn.load_state_dict(tcn_sd)
    return enc, tcn
A, Bm = mk("bf16"), mk("fp32")
xa = synth.synth_audio((3, 2, 40000), seed=300).to(dev)
emb = synth.synth_audio((1, 2048), seed=9).to(dev)
M1 = torch.randn(4096, 4096, device=dev)
A[1](xa, emb); _lib.lib().check(_lib.lib().mst_tcn_set_tuning(A[1]._handle, 1), "t")
dist = {"none": lambda: None, "bf16 encoder": lambda: A[0](xa), "bf16 tcn one-tile forms": lambda: A[1](xa, emb)}
for kb in (8, 16, 40):
    for dname, dfn in dist.items():
        dfn(); torch.cuda.synchronize()
        out = torch.zeros(4 + 4 * 200, dtype=torch.int32, device=dev)
        stop = [False]
        def ta():
            with torch.cuda.stream(torch.cuda.Stream(device=dev)):
                while not stop[0]:
                    for _ in range(10):
                        dfn()
                    torch.cuda.current_stream().synchronize()
        def tb():
            s = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(s):
                for rep in range(6):
                    probe.lds_probe_launch(C.c_void_p(s.cuda_stream), 2048, 200, C.c_void_p(out.data_ptr()), kb)
                s.synchronize()
                stop[0] = True
        ths = [threading.Thread(target=ta), threading.Thread(target=tb)]
        [t.start() for t in ths]; [t.join() for t in ths]
        o = out.cpu().numpy().view("uint32")
        first = [(int(o[4 + 4 * i]), int(o[5 + 4 * i]), hex(int(o[6 + 4 * i])), int(o[7 + 4 * i])) for i in range(min(int(o[0]), 6))]
        print(f"probe {kb:2d} KB LDS | disturber {dname:24s}: LDS words changed {int(o[0])}, register mismatches {int(o[1])}, wide-access mismatches {int(o[2])}; first (wg, word, value, iter): {first}", flush=True)

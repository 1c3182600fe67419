import cProfile, pstats, sys, os, io
sys.argv = ["bench_cli.py", "--precision", "bf16"]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_cli
import music_mixing_style_transfer_amd.inference.style_transfer as st
orig = st.Mixing_Style_Transfer_Inference.inference
calls = {"n": 0}
def wrapped(self):
    calls["n"] += 1
    if calls["n"] == 2:
        pr = cProfile.Profile(); pr.enable(); r = orig(self); 
        import torch; torch.cuda.synchronize(); pr.disable()
        s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])
        return r
    return orig(self)
st.Mixing_Style_Transfer_Inference.inference = wrapped
bench_cli.main()

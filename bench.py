#!/usr/bin/env python
"""bench.py - headline benchmark: stereo 44.1 kHz segments/sec, FXencoder + MixFXcloner forward.

Headline workload (BASELINE.json configs[1]): batch = 32 segments of 2 x 131072 samples per GPU, default configs.yaml
architectures, deterministic synthetic weights, synthetic audio; one step = FXencoder on the 32 reference segments ->
(all-gather of segment embeddings when N > 1) -> mean -> FiLM factors -> TCN on the 32 input segments.  Inputs are
resident in HBM before the timed region.  Weak scaling: every rank owns 32 segments.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--precision bf16|fp32] [--workload all|configs1|track60]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line (rank 0), kept under 4 KB so that it survives any log tail: per leg only value / ms / frac / max_abs_vs_oracle.
Everything else (per-block times, thread sweeps, workload prose, the normaliser's per-effect times) goes to the DETAILS file the line names
("details": gpurun_out/bench_details.json, also copied to profiles/ by tools/gpu_round.sh).  Objects carried by the line:
  "roofline"      dominant kernel = the dilated 128x128x15 TCN block conv, timed with HIP events on its stream inside
                  the timed region;
  "track60"       BASELINE configs[4]: ONE 60-minute stereo stem (158 760 000 samples = 1212 segments of 131072 for the
                  reference and the input role) through StyleTransferEngine.transfer_stem, the 1212 segments split over
                  the N ranks (STRONG scaling; the only collective is the all-gather of segment embeddings).  Reported with
                  the stems resident in HBM ("value") and host-to-host from / to pinned memory ("pcie_inclusive": per-rank
                  shard H2D + D2H overlapped with the networks).  N > 1 also times the same track on rank 0 alone
                  ("t1_ms_same_job") so that T1 / (N * TN) can be read off one line;
  "parity_mode"   (N = 1) the same configs[1] step in the exact-fp32 mode that meets north_star's 1e-4 tolerance, its
                  dominant kernel against the fp32 MFMA peak, and the max-abs deviation of the HIP path from the oracle;
  "bf16x3_mode"   (N = 1) the same step in the split-bf16 mode (three bf16 MFMAs per product): <= 1e-4 as well, at bf16-MFMA speed / 3;
  "fx_chain"      (N = 1) BASELINE configs[3]: EQ -> rms -> compressor -> rms -> imager -> rms -> gain on 64 segments;
  "cpu_baseline"  (N = 1) the oracle (torch-CPU restatement of the reference) on a bounded sample, all cores and 1 thread.
With --workload track60 the track IS the headline ("scaling": "strong").
"""
import argparse
import ctypes as C
import json
import os
import statistics
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

SEG_LEN = 131072
BATCH = 32
TRACK_SAMPLES = 60 * 60 * 44100                         # 158 760 000 -> 1211 full segments + a zero-padded tail
TCN_FLOP_PER_SAMPLE_BLOCK = 2 * 128 * 128 * 15          # one dense TCN block, per output time step
PEAK = {"bf16": 2500.0, "fp32": 157.3, "bf16x3": 2500.0}     # dense MFMA TFLOP/s, MI355X_MICROARCH.md
MFMA_PER_FLOP = {"bf16": 1, "fp32": 1, "bf16x3": 3}         # bf16x3 issues three bf16 MFMAs per algorithmic product
HBM_PEAK_GBPS = 8000.0


DETAILS_PATH = os.path.join(REPO, "gpurun_out", "bench_details.json")


class BoxSampler:
    """Shader clock and socket power of the GPU while a timed region runs: a thread polls the amdgpu hwmon files of the device
    (freq1_input in Hz, power1_input in microwatts) every 20 ms (~20 samples per timed region: the poll shares the
    interpreter lock with the thread that enqueues the steps, so it is kept rare).  None when the files cannot be found."""

    def __init__(self, dev_index, period_s=0.02):
        import glob
        self.files = None
        try:
            pr = torch.cuda.get_device_properties(dev_index)
            bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            hw = glob.glob(f"/sys/bus/pci/devices/{bdf}/hwmon/hwmon*")
            if hw and os.path.exists(os.path.join(hw[0], "freq1_input")):
                self.files = (os.path.join(hw[0], "freq1_input"), os.path.join(hw[0], "power1_input"))
        except Exception:
            self.files = None
        self.period, self.sclk, self.power, self._stop, self._thr = period_s, [], [], None, None

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return float(f.read().strip())
        except Exception:
            return None

    def __enter__(self):
        if self.files is None:
            return self
        import threading
        self._stop = threading.Event()

        def poll():
            while not self._stop.is_set():
                f, p = self._read(self.files[0]), self._read(self.files[1])
                if f:
                    self.sclk.append(f / 1e6)
                if p:
                    self.power.append(p / 1e6)
                time.sleep(self.period)
        self._thr = threading.Thread(target=poll, daemon=True)
        self._thr.start()
        return self

    def __exit__(self, *exc):
        if self._thr is not None:
            self._stop.set()
            self._thr.join(timeout=1.0)

    def summary(self):
        if not self.sclk:
            return {"sclk_mhz": None, "power_w": None}
        return {"sclk_mhz": round(statistics.median(self.sclk)), "sclk_mhz_min": round(min(self.sclk)), "samples": len(self.sclk),
                "power_w": round(statistics.fmean(self.power)) if self.power else None}


def calibrate(lib, dev, launches=24):
    """The bare main loop of the bf16 block kernel on this box (mst_calib_mainloop): (ms per launch-equivalent, shader MHz inside it)."""
    ms, mhz = C.c_float(0), C.c_float(0)
    with torch.cuda.device(dev):
        lib.check(lib.mst_calib_mainloop(launches, C.byref(ms), C.byref(mhz), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
                  "mst_calib_mainloop")
    return float(ms.value), float(mhz.value)


def load_cfg():
    import yaml
    with open(os.path.join(REPO, "music_mixing_style_transfer_amd", "networks", "configs.yaml")) as f:
        c = yaml.full_load(f)
    return c["Effects_Encoder"]["default"], c["TCN"]["default"]


# ------------------------------------------------------------------------------------------------ CPU baseline (oracle)
def cpu_baseline(enc_cfg, enc_sd, tcn_sd):
    """Oracle (torch-CPU fp32 restatement of the reference) on a bounded sample of the configs[1] workload: one warm-up
    + the median of 3 segments on all cores; one thread on an eighth of a segment (the TCN is linear in the length)."""
    from oracle import networks_ref as R
    from music_mixing_style_transfer_amd.utils import synth
    cfg = dict(enc_cfg)

    keep = {}

    def one(x):
        t0 = time.perf_counter()
        emb = R.fxencoder_forward(enc_sd, cfg, x)
        y = R.tcn_forward(tcn_sd, x, emb.mean(0, keepdim=True))
        dt = time.perf_counter() - t0
        keep[x.shape[-1]] = y
        return dt

    all_cores = torch.get_num_threads()
    x = synth.synth_audio((1, 2, SEG_LEN), seed=1)
    one(x)                                                           # warm-up (allocator, oneDNN primitive cache)
    sweep = {}
    for nt in sorted({n for n in (8, 16, 32, 64, all_cores) if n <= all_cores}):      # torch's CPU convs do not scale to 128 threads
        torch.set_num_threads(nt)
        sweep[nt] = one(x)
    cores = min(sweep, key=sweep.get)
    torch.set_num_threads(cores)
    ts = [one(x) for _ in range(3)]
    xs = synth.synth_audio((1, 2, SEG_LEN // 8), seed=1)
    torch.set_num_threads(1)
    try:
        one(xs)
        t1 = statistics.median([one(xs) for _ in range(2)]) * 8.0
    finally:
        torch.set_num_threads(all_cores)
    return {"y_ref": keep[SEG_LEN], "x": x,
            "value": 1.0 / statistics.median(ts), "unit": "segments/s", "cores": cores, "kind": "port",
            "samples_s": [round(t, 3) for t in ts], "thread_sweep_s": {str(k): round(v, 3) for k, v in sweep.items()},
            "value_all_cores": 1.0 / sweep[all_cores], "host_cores": all_cores, "value_1thread": 1.0 / t1,
            "sample": f"FXencoder+TCN fp32 forward of 1 segment of 2x{SEG_LEN} (oracle/networks_ref.py, torch-CPU): one warm-up, one sample "
                      f"per thread count of the sweep, then the warm median of 3 at the fastest count ({cores} threads); 1 thread: "
                      f"median of 2 x 1/8 segment (2x{SEG_LEN // 8}) scaled by 8"}


# ------------------------------------------------------------------------------------------------ configs[1] step
def bench_configs1(engine, tcn, lib, ref, inp, steps, warmup, world, dist, dev, stats=None):
    """K timed steps between barriers; returns (seconds max over ranks, per-block kernel ms, forwards timed).  stats (a dict) also
    receives the per-step durations (HIP events on the stream the steps run on) and the clock / power samples of the region."""
    for _ in range(warmup):
        engine.step(ref, inp)
    torch.cuda.synchronize()
    lib.check(lib.mst_tcn_timing_begin(tcn._handle, steps), "timing_begin")
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)] if stats is not None else None
    sampler = BoxSampler(dev.index if dev.index is not None else 0) if stats is not None else None
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    if sampler is not None:
        sampler.__enter__()
    t0 = time.perf_counter()
    for k in range(steps):
        if marks is not None:
            marks[k].record()
        engine.step(ref, inp)
    if marks is not None:
        marks[steps].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if sampler is not None:
        sampler.__exit__()
        per = [marks[k].elapsed_time(marks[k + 1]) for k in range(steps)]
        stats["step_ms"] = {"median": statistics.median(per), "min": min(per), "max": max(per)}
        stats["box"] = sampler.summary()
    nb = tcn.hparams.nblocks
    ms = (C.c_float * (nb + 1))()
    nf = C.c_int(0)
    lib.check(lib.mst_tcn_timing_end(tcn._handle, ms, C.byref(nf)), "timing_end")
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    if stats is not None:
        stats["fused0"] = tcn_fused_block0(lib, tcn)
    return float(tmax.item()), [float(v) for v in ms], int(nf.value)


KERNEL_SOURCES = {"tcn": ("tcn_kernels.h", "mst_dev.h", "mst_rt.h"), "fx": ("fx_kernels.h", "fft_kernels.h", "mst_dev.h", "mst_rt.h")}


def csrc_sha256(family):
    """Fingerprint of the kernel sources of one family ("tcn": the block kernels, "fx": the FX processors).  The counter files under
    profiles/ carry the fingerprint of the sources they were measured on (tools/pmc_traffic.py, tools/pmc_fx_traffic.py write it), and a
    stored `traffic` figure of other kernel sources is not printed.  (Sources, not the binary: the .so is rebuilt per checkout.)"""
    import hashlib
    h = hashlib.sha256()
    for name in KERNEL_SOURCES[family]:
        with open(os.path.join(REPO, "music_mixing_style_transfer_amd", "csrc", name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()


def stored_traffic(name, family):
    """(traffic bytes, source note) from profiles/<name> if that file was measured on THESE kernel sources, else (None, why)."""
    tpath = os.path.join(REPO, "profiles", name)
    if not os.path.exists(tpath):
        return None, f"profiles/{name} absent"
    with open(tpath) as f:
        rec = json.load(f)
    if rec.get("csrc_sha256") != csrc_sha256(family):
        return None, f"profiles/{name} was measured on other kernel sources (csrc_sha256 differs): not printed"
    return rec.get("traffic_bytes"), f"profiles/{name} (offline rocprofv3 --pmc passes, same kernel sources)"


def tcn_fused_block0(lib, tcn):
    """Whether the handle's last forward ran block 0 inside block 1's launch (mst_tcn_get_tuning; bit 5 is a request with conditions)."""
    fl, fused = C.c_int(0), C.c_int(0)
    lib.check(lib.mst_tcn_get_tuning(tcn._handle, C.byref(fl), C.byref(fused)), "mst_tcn_get_tuning")
    return bool(fused.value)


def roofline(block_ms, nb, B, precision, traffic=None, fused0=False):
    dense = block_ms[1:nb]                                   # the 13 dilated 128->128 blocks
    if fused0:          # mst_tcn_set_tuning bit 5 applied: block 0 ran inside block 1's launch (no launch of its own)
        dense = block_ms[2:nb]                               # that launch also does block 0's work: not a plain dense launch
    avg_ms = sum(dense) / len(dense)
    flop = TCN_FLOP_PER_SAMPLE_BLOCK * B * SEG_LEN
    achieved = flop / (avg_ms * 1e-3) / 1e12
    out = {"kernel": "%s (dilated 128x128x15 conv + fused BN/LeakyReLU/FiLM/residual)" %
                     {"bf16": "tcn_block_bf16_kernel<4, false, 8, 2> (9 of the 13 launches, d = 4 ... 1024: four-phase 256-time tiles, class-major main loop, one tile per workgroup, two workgroups per CU; d = 2048 / 4096 / 8192 - one 256-time tile = a whole phase sequence of 4 x 64 / 8 x 32 / 16 x 16 steps - the unrolled forms <4 | 8 | 16, ., 8, 1> of the same kernel without all-padding tap tiles, the last with the fused head; d = 2 + block 0 in one launch of its two-phase form <2, false, 8, 2, true>)",
                      "fp32": "tcn_block_f32_kernel", "bf16x3": "tcn_block_bf16x3_kernel / _half_kernel"}[precision],
           "bound": "mfma", "achieved": achieved, "peak": PEAK[precision], "unit": "TFLOP/s",
           "frac": achieved / PEAK[precision], "traffic": traffic, "avg_launch_ms": avg_ms, "launches_per_step": len(dense),
           "flop_per_launch": flop, "per_block_ms": block_ms}
    if precision == "bf16" and nb == 14 and SEG_LEN == 131072 and len(block_ms) >= 14:
        # the nine launches d = 4 ... 1024 run the generic tile form and execute every one of the 2.06 TFLOP; d = 2048 / 4096 / 8192 run whole-sequence tiles
        # that leave out the multiplications by zero padding (3 % / 10 % / 23 % of their MFMAs) - their share of `frac` is algorithmic work not executed
        gen = block_ms[2:11]                                 # blocks 2 ... 10 = d = 4 ... 1024
        out["generic_launch_ms"] = sum(gen) / len(gen)
        out["frac_generic_launches"] = flop / (out["generic_launch_ms"] * 1e-3) / 1e12 / PEAK[precision]
    if MFMA_PER_FLOP[precision] > 1:      # `achieved` / `frac` count ALGORITHMIC flops; the matrix pipe executes three times as many
        out["mfma_flop_per_algorithmic_flop"] = MFMA_PER_FLOP[precision]
        out["mfma_pipe_frac"] = MFMA_PER_FLOP[precision] * achieved / PEAK[precision]
    return out


def slim_roofline(rl):
    """The roofline object of the printed line: numbers only (kernel names shortened, per-block times in the details file)."""
    keep = ("bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "launches_per_step", "calib_ms", "frac_of_box_mainloop",
            "calib_sclk_mhz", "sclk_mhz", "power_w", "traffic_source", "generic_launch_ms", "frac_generic_launches")
    out = {"kernel": rl["kernel"].split(" (")[0]}
    for k in keep:
        if k in rl and rl[k] is not None:
            v = rl[k]
            out[k] = round(v, 4) if isinstance(v, float) and k not in ("traffic",) else v
    if "traffic" not in out:
        out["traffic"] = None
    return out


# ------------------------------------------------------------------------------------------------ 60-minute track
def bench_track60(enc, tcn, world, rank, dist, dev, steps, warmup, with_host=True, with_t1=True):
    """One 60-minute stem (reference + input role), its 1212 segments split over the ranks."""
    from music_mixing_style_transfer_amd.inference import StyleTransferEngine
    from music_mixing_style_transfer_amd.inference import segmentation as seg
    g = torch.Generator(device=dev).manual_seed(1234)             # the same track on every rank
    x_in = (torch.rand(2, TRACK_SAMPLES, generator=g, device=dev) * 2 - 1)
    x_ref = (torch.rand(2, TRACK_SAMPLES, generator=g, device=dev) * 2 - 1)
    n_seg = seg.segment_count(TRACK_SAMPLES, SEG_LEN)

    def timed(engine, a, b, k, w, sync_ranks):
        for _ in range(w):
            engine.transfer_stem(a, b, SEG_LEN, SEG_LEN)
        torch.cuda.synchronize()
        if sync_ranks:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            engine.transfer_stem(a, b, SEG_LEN, SEG_LEN)
        torch.cuda.synchronize()
        if sync_ranks:
            dist.barrier()
        dt = (time.perf_counter() - t0) / k
        if sync_ranks:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    eng = StyleTransferEngine(enc, tcn)
    t_res = timed(eng, x_in, x_ref, steps, warmup, world > 1)
    lo, hi = seg.shard_range(n_seg, rank, world)
    out = {"workload": f"configs[4]: one 60-min stereo stem = {n_seg} segments of 2x{SEG_LEN} (reference and input role), "
                       f"contiguous segment shards over {world} GPU(s), all-gather of [{n_seg}, 2048] embeddings, "
                       f"passes of <= {eng.pass_samples // SEG_LEN} segments",
           "scaling": "strong", "segments": n_seg, "segments_rank0": hi - lo, "segments_rank_local": hi - lo, "n_gpus": world,
           "value": n_seg / t_res, "unit": "segments/s", "t_ms": t_res * 1e3}
    if with_host:
        h_in, h_ref = x_in.cpu().pin_memory(), x_ref.cpu().pin_memory()
        t_host = timed(eng, h_in, h_ref, max(1, steps // 2), 1, world > 1)
        out["pcie_inclusive"] = {"value": n_seg / t_host, "unit": "segments/s", "t_ms": t_host * 1e3,
                                 "what": "stems in pinned host memory -> each rank uploads / converts / downloads only its "
                                         "shard (copy streams overlap the networks) -> converted shard in pinned host memory"}
        del h_in, h_ref
    if world > 1 and with_t1:
        if rank == 0:
            solo = StyleTransferEngine(enc, tcn)
            solo.dist, solo.world, solo.rank = None, 1, 0          # the same track on this GPU alone
            t1 = timed(solo, x_in, x_ref, 1, 1, False)
            out["t1_ms_same_job"] = t1 * 1e3
            out["efficiency_t1_over_n_tn"] = t1 / (world * t_res)
        dist.barrier()
    return out


# ------------------------------------------------------------------------------------------------ parity mode (fp32)
def bench_parity_mode(engine, enc, tcn, lib, ref, inp, dev, enc_cfg, enc_sd, tcn_sd, precision="fp32"):
    from music_mixing_style_transfer_amd.utils import synth
    from oracle import networks_ref as R
    enc.precision = tcn.precision = precision
    try:
        nsteps = 2 if precision == "fp32" else 4
        dt, block_ms, _ = bench_configs1(engine, tcn, lib, ref, inp, nsteps, 2, 1, None, dev)
        fused0 = tcn_fused_block0(lib, tcn)
        L = 16384
        pr, pi = synth.synth_audio((2, 2, L), seed=5), synth.synth_audio((2, 2, L), seed=6)
        y, _ = engine.step(pr.to(dev), pi.to(dev))
        _, _, y_ref = R.style_transfer_segments(enc_sd, dict(enc_cfg), tcn_sd, pr, pi)
        err = float((y.cpu() - y_ref).abs().max())
    finally:
        enc.precision = tcn.precision = "bf16"
    return {"dtype": "f32" if precision == "fp32" else "bf16x3 (fp32 operands split hi + lo, three bf16 MFMAs per product, fp32 accumulate; FXencoder and TCN)",
            "value": BATCH * nsteps / dt, "unit": "segments/s", "ms_per_step": dt / nsteps * 1e3, "steps": nsteps,
            "roofline": roofline(block_ms, tcn.hparams.nblocks, BATCH, precision, fused0=fused0),
            "max_abs_vs_oracle": err, "probe": f"2 reference + 2 input segments of 2x{L} vs oracle/networks_ref.py", "tolerance": 1e-4}


# ------------------------------------------------------------------------------------------------ FX chain (config 4)
def bench_input_normalizer():
    """Row F: Audio_Effects_Normalizer (the reference CLI's default --normalize_input True) on two 3-minute stereo stems, the oracle
    chain on a 6 s excerpt beside it (tools/bench_normalizer.py)."""
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import bench_normalizer
    r = bench_normalizer.run(180.0, 6.0, "drums,other")
    return {"workload": "loudness -> eq -> compression -> imager -> loudness on 2 stems of [7938000, 2] (host in, host out)",
            "value": r["value"], "unit": r["unit"], "s_per_stem": r["s_per_stem"], "s_per_effect": r["s_per_effect"],
            "rel_dev_vs_oracle_on_excerpt": r["rel_dev_vs_oracle_on_excerpt"], "cpu_baseline": r["cpu_baseline"],
            "parity": "unpinned for the BS.1770 meter (pyloudnorm) and the onset detector (aubio): both sides of the excerpt comparison are "
                      "restatements of the published algorithms; the imager step incl. its Haas branch, the compressor and the matching "
                      "glue are pinned by the reference's own outputs (tests/golden/normalizer.npz)"}


def bench_fx_chain(dev, steps=5):
    import numpy as np
    from music_mixing_style_transfer_amd.mixing_manipulator import (AugmentationChain, Compressor, Equaliser, Gain,
                                                                    MidSideImager)
    from oracle import fx_ref as F
    n, L = 64, SEG_LEN
    g = torch.Generator().manual_seed(0)
    x = (0.1 * torch.randn(n, L, 2, generator=g)).clamp_(-1, 1).to(dev)
    eq = Equaliser(2, 44100)
    for band, (gg, _, _) in F.CONFIG4["eq"].items():
        getattr(eq.parameters, band + "_gain").value = gg
    comp, im, gn = Compressor(44100), MidSideImager(), Gain()
    for k, v in F.CONFIG4["comp"].items():
        getattr(comp.parameters, k).value = v
    im.parameters.bal.value = F.CONFIG4["imager_bal"]
    gn.parameters.gain.value = F.CONFIG4["gain_db"]
    chain = AugmentationChain(fxs=[(eq, 1.0, True), (comp, 1.0, True), (im, 1.0, True), (gn, 1.0, False)],
                              randomize_param_value=False)
    out = chain([x])[0]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = chain([x])[0]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    x17, c_comp = x[17].cpu().numpy(), _oracle_c_compressor()
    ref = F.fx_chain(x17, compressor_fn=c_comp)
    dev_max = float(np.abs(out[17].cpu().numpy() - ref).max())
    # the oracle chain (numpy / scipy float64 + the C compressor of oracle/fx_ref.c) on this host, ONE thread, a bounded sample: median of 5 segments
    ts = []
    for k in range(5):
        xk = x[20 + k].cpu().numpy()
        tc = time.perf_counter()
        F.fx_chain(xk, compressor_fn=c_comp)
        ts.append(time.perf_counter() - tc)
    cpu = {"value": 1.0 / statistics.median(ts), "unit": "segments/s", "cores": 1, "kind": "port",
           "sample": "oracle/fx_ref.py chain (scipy lfilter EQ, oracle/fx_ref.c compressor, numpy imager / gain / rms) on 5 segments of [131072, 2], median, 1 thread"}
    # row f-3 beside it: the convolution reverb (1.5 s stereo impulse response) on the same batch - the library's own FFT kernels
    from music_mixing_style_transfer_amd.mixing_manipulator import ConvolutionalReverb
    from music_mixing_style_transfer_amd.utils import synth
    Lh = 66150
    hl = (synth.synth_audio((Lh, 2), seed=9).numpy().astype(np.float64) * np.exp(-np.arange(Lh) / 12000.0)[:, None] * 0.05).astype(np.float32)
    hl[441] = (0.8, 0.7)
    rv = ConvolutionalReverb([[{"impulse_response": (lambda: hl)}]], 44100)
    rv.update()
    yr = rv.process(x)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(steps):
        yr = rv.process(x)
    torch.cuda.synchronize()
    rv_dt = (time.perf_counter() - t1) / steps
    rv_ref = F.conv_reverb(x[3].cpu().numpy(), hl)
    rv_dev = float(np.abs(yr[3].cpu().numpy() - rv_ref).max() / np.abs(rv_ref).max())
    alg = 144 * L * n                       # unfused per-processor read + write bytes of the chain
    # what the chain REALLY moves: FETCH_SIZE / WRITE_SIZE counter passes over its kernels (tools/gpu_fx_pmc.sh -> profiles/*fx_chain_traffic.json;
    # offline: counters need their own rocprofv3 passes).  `achieved` / `frac` stay on the ALGORITHMIC basis (one read + one write of the audio,
    # 16 L per segment - what a perfectly fused chain would move); `frac_on_traffic` is the measured bytes over the same time.
    traffic, tsrc = stored_traffic("r06_fx_chain_traffic.json", "fx")
    return {"workload": "configs[3]: EQ -> rms -> compressor -> rms -> imager -> rms -> gain on 64 segments of [131072, 2]",
            "value": n / dt, "unit": "segments/s", "ms_per_chain": dt * 1e3,
            "roofline": {"bound": "hbm", "achieved": 16 * L * n / dt / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": 16 * L * n / dt / 1e9 / HBM_PEAK_GBPS, "algorithmic_bytes": 16 * L * n,
                         "bytes_basis": "one read + one write of [n, L, 2] float32 (a perfectly fused chain)",
                         "traffic": traffic, "traffic_source": tsrc,
                         "traffic_over_algorithmic": round(traffic / (16 * L * n), 2) if traffic else None,
                         "frac_on_traffic": round(traffic / dt / 1e9 / HBM_PEAK_GBPS, 4) if traffic else None,
                         "equivalent_unfused_GBps": alg / dt / 1e9, "unfused_bytes_survey_8d": alg},
            "max_abs_vs_oracle": dev_max, "probe": "item 17 vs oracle/fx_ref.py chain (EQ parity unpinned, see DESIGN.md)",
            "tolerance": "2e-6 * max|ref|", "cpu_baseline": cpu,
            "conv_reverb": {"value": n / rv_dt, "unit": "segments/s", "ms_per_batch": rv_dt * 1e3, "ir_samples": Lh,
                            "rel_dev_vs_oracle": rv_dev, "what": "ConvolutionalReverb.process on the same 64 segments, 1.5 s stereo response"}}


def _oracle_c_compressor():
    import subprocess
    import numpy as np
    subprocess.run(["make", "-C", os.path.join(REPO, "oracle")], check=True, capture_output=True)
    lib = C.CDLL(os.path.join(REPO, "oracle", "libfx_ref.so"))
    fp = C.POINTER(C.c_float)

    def c_comp(xx, threshold, attack_time, release_time, ratio, sample_rate):
        xx = np.ascontiguousarray(xx, dtype=np.float32)
        yy = np.empty_like(xx)
        lib.ref_compressor(xx.ctypes.data_as(fp), yy.ctypes.data_as(fp), C.c_long(xx.shape[0]), xx.shape[1],
                           C.c_double(threshold), C.c_double(attack_time), C.c_double(release_time), C.c_double(ratio),
                           C.c_double(0.0), C.c_double(sample_rate))
        return yy
    return c_comp


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32", "bf16x3"])
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--workload", default="all", choices=["all", "configs1", "track60"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--x3-large-tiles", action="store_true", help="bf16x3: 256-time tiles, one workgroup per CU (mst_tcn_set_tuning bit 0 off)")
    ap.add_argument("--enc-schedule", type=int, default=None, help="mst_enc_set_schedule flags (bit 0: weight-major workgroup order of the weight-heavy encoder layers, bit 1: 2 x 2 wave tiling of the 128-channel conv kernel)")
    ap.add_argument("--enc-rows-min-tiles", type=int, default=None, help="mst_enc_set_tuning: tiles from which a layer keeps its rows resident in LDS (-1 never)")
    ap.add_argument("--tcn-tuning", type=int, default=None, help="mst_tcn_set_tuning flags (bit 0: bf16x3 small tiles, bits 1-2: form of the bf16 block kernel, include/mst_hip.h)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` typed like the N = 1 command: start the N ranks ourselves (one process per GPU through
        # torch.distributed.run on 127.0.0.1 with a free port); rank 0 of that job prints the ONE line to our stdout
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "8")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd, env=env).returncode)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was started inside a job of WORLD_SIZE={world}: the two must agree "
                         f"(torch.distributed.run --nproc-per-node {args.gpus}, or no launcher at all)")
    import torch.distributed as dist
    if os.environ.get("MST_BENCH_SHARE_GPU"):      # test hook: several ranks on one GPU (with MST_DIST_BACKEND=gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("MST_DIST_BACKEND", "nccl")          # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
        seen = torch.ones(1, dtype=torch.int64, device=dev if backend == "nccl" else "cpu")      # every rank really takes part in a collective
        dist.all_reduce(seen)
        ranks_seen = int(seen.item())
        assert ranks_seen == args.gpus, (ranks_seen, args.gpus)
        comm = {"backend": backend}
        if backend == "nccl":
            try:
                comm["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception as e:          # the build of torch decides whether the query exists; the run does not depend on it
                comm["rccl_version"] = f"unknown ({type(e).__name__})"
    else:
        ranks_seen, comm = 1, None

    from music_mixing_style_transfer_amd import _lib
    from music_mixing_style_transfer_amd.inference import StyleTransferEngine, build_models
    from music_mixing_style_transfer_amd.utils import synth

    enc_cfg, tcn_cfg = load_cfg()
    enc_sd = synth.fxencoder_state_dict(enc_cfg, seed=0)
    tcn_sd = synth.tcn_state_dict(seed=0)
    enc, tcn = build_models({k: (list(v) if isinstance(v, list) else v) for k, v in enc_cfg.items()}, tcn_cfg, dev,
                            precision=args.precision)
    enc.load_state_dict(enc_sd)
    tcn.load_state_dict(tcn_sd)
    engine = StyleTransferEngine(enc, tcn)
    lib = _lib.lib()
    enc._get_runner()._ensure(lib)      # weight folding / packing is setup, not part of a step
    tcn._ensure(lib)
    if args.enc_schedule is not None:
        lib.check(lib.mst_enc_set_schedule(enc._get_runner().handle, args.enc_schedule), "mst_enc_set_schedule")
    if args.enc_rows_min_tiles is not None:
        lib.check(lib.mst_enc_set_tuning(enc._get_runner().handle, args.enc_rows_min_tiles), "mst_enc_set_tuning")
    if args.x3_large_tiles:
        lib.check(lib.mst_tcn_set_tuning(tcn._handle, _lib.TCN_TUNING_DEFAULT & ~1), "mst_tcn_set_tuning")      # bit 0 off, the bf16 form unchanged
    if args.tcn_tuning is not None:
        lib.check(lib.mst_tcn_set_tuning(tcn._handle, args.tcn_tuning), "mst_tcn_set_tuning")
    nb = tcn.hparams.nblocks
    B = args.batch
    dtype = {"bf16": "bf16", "fp32": "f32", "bf16x3": "bf16x3"}[args.precision]
    base = {"n_gpus": world, "ranks_seen": ranks_seen, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "vs_baseline": None,
            "dtype": dtype, "data": "synthetic"}
    if comm is not None:
        base["comm"] = comm

    def per_rank(n):
        """[n of rank 0, n of rank 1, ...]: what every rank says it owns, gathered with the job's own backend (proof that N ranks worked)"""
        if world == 1:
            return [int(n)]
        t = torch.tensor([int(n)], dtype=torch.int64, device=dev if comm["backend"] == "nccl" else "cpu")
        got = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(got, t)
        return [int(g.item()) for g in got]

    if args.workload == "track60":
        tr = bench_track60(enc, tcn, world, rank, dist, dev, args.steps, args.warmup)
        tr["segments_rank"] = per_rank(tr["segments_rank_local"])
        if rank == 0:
            out = dict(base, metric="stereo 44.1 kHz segments/sec (FXencoder+MixFXcloner fwd)", value=tr["value"],
                       unit="segments/s", ms_per_step=tr["t_ms"], scaling="strong",
                       config={"workload": tr["workload"], "segment_length": SEG_LEN, "segments": tr["segments"],
                               "parallelism": f"segment-sharded x{world}, all-gather of embeddings"}, track60=tr)
            print(json.dumps(out), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return

    ref = synth.synth_audio((B, 2, SEG_LEN), seed=100 + rank).to(dev)      # resident in HBM before timing
    inp = synth.synth_audio((B, 2, SEG_LEN), seed=200 + rank).to(dev)
    stats = {}
    dt, block_ms, nf = bench_configs1(engine, tcn, lib, ref, inp, args.steps, args.warmup, world, dist, dev, stats)
    calib_ms, calib_mhz = calibrate(lib, dev) if args.precision == "bf16" else (None, None)      # right behind the timed region: the chip is warm
    track = None
    if args.workload == "all" and args.precision == "bf16":
        track = bench_track60(enc, tcn, world, rank, dist, dev, 2, 1)
        track["segments_rank"] = per_rank(track["segments_rank_local"])
    seg_rank = per_rank(B)

    if rank == 0:
        # HBM bytes per launch of the dominant kernel: PMC counters need their own rocprofv3 passes (tools/pmc_traffic.py), so the figure is a
        # stored one - printed only when it was measured on this build of the kernels and with the default tuning flags, else null + the reason
        traffic, tsrc = (None, "counter file covers the bf16 headline workload with the default tuning flags only")
        if args.precision == "bf16" and B == BATCH and args.tcn_tuning in (None, _lib.TCN_TUNING_DEFAULT):
            traffic, tsrc = stored_traffic("r06_tcn_block_bf16_traffic.json", "tcn")
        rl = roofline(block_ms, nb, B, args.precision, traffic, fused0=stats.get("fused0", False))
        rl["timed_forwards"] = nf
        rl["traffic_source"] = tsrc
        rl["block0_fused_into_block1"] = stats.get("fused0", False)
        if calib_ms is not None:
            # the box: the bare main loop of the block kernel (same arithmetic as one dense launch) timed in this process, the shader
            # clock inside it, and clock / power sampled from the driver while the timed steps ran
            rl["calib_ms"] = calib_ms
            rl["frac_of_box_mainloop"] = calib_ms / rl["avg_launch_ms"]
            rl["calib_sclk_mhz"] = round(calib_mhz)
        rl.update(stats.get("box", {}))
        details = {"headline": {"roofline": dict(rl), "step_ms": stats.get("step_ms")}}
        workload = (f"configs[1]: batch={B} segments of 2x{SEG_LEN} per GPU, FXencoder+MixFXcloner fwd, default configs.yaml nets, "
                    f"synthetic weights, {args.precision} MFMA (fp32 accumulate)")
        out = dict(base, metric="stereo 44.1 kHz segments/sec (FXencoder+MixFXcloner fwd)",
                   value=world * B * args.steps / dt, unit="segments/s", ms_per_step=dt / args.steps * 1e3, scaling="weak",
                   config={"workload": workload, "segments_per_gpu": B, "segment_length": SEG_LEN, "segments_rank": seg_rank,
                           "parallelism": f"segment-sharded x{world}, all-gather of embeddings"},
                   step_ms={k: round(v, 3) for k, v in (stats.get("step_ms") or {}).items()},
                   roofline=slim_roofline(rl))
        if world == 1 and args.precision in ("bf16", "bf16x3"):
            # row f-1 (inference/feature_extraction.py): the FXencoder alone on the same 32 resident segments - its segments/s is the whole cost
            # of the feature-extraction CLI's network part
            for _ in range(2):
                enc(ref)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                enc(ref)
            torch.cuda.synchronize()
            fe_ms = (time.perf_counter() - t0) / 10 * 1e3
            out["feature_extraction"] = {"value": round(B / fe_ms * 1e3, 1), "unit": "segments/s", "ms": round(fe_ms, 3),
                                         "what": "FXencoder forward only, same batch"}
            details["feature_extraction"] = dict(out["feature_extraction"], flop_per_batch=457.6e9 * B / 32,
                                                 frac_of_bf16_peak=457.6e9 * B / 32 / (fe_ms * 1e-3) / 1e12 / PEAK["bf16"] * MFMA_PER_FLOP[args.precision])
        if track is not None:
            details["track60"] = track
            out["track60"] = {"value": round(track["value"], 1), "ms": round(track["t_ms"], 1), "segments": track["segments"], "scaling": "strong",
                              "segments_rank": track["segments_rank"]}
            if "pcie_inclusive" in track:
                out["track60"]["pcie_value"] = round(track["pcie_inclusive"]["value"], 1)
            for k in ("t1_ms_same_job", "efficiency_t1_over_n_tn"):
                if k in track:
                    out["track60"][k] = round(track[k], 4)
        if world == 1 and args.workload == "all" and args.precision == "bf16" and B == BATCH:
            sys.path.insert(0, os.path.join(REPO, "tools"))
            import bench_cli
            import contextlib
            with contextlib.redirect_stdout(sys.stderr):      # the runner prints its progress like the reference CLI: keep stdout to the ONE line
                f2f = bench_cli.run(180.0, "bf16", songs=2)      # wav to wav, second pass over two 3-minute songs
            details["file_to_file"] = f2f
            for key, prec in (("parity_mode", "fp32"), ("bf16x3_mode", "bf16x3")):
                leg = bench_parity_mode(engine, enc, tcn, lib, ref, inp, dev, enc_cfg, enc_sd, tcn_sd, prec)
                details[key] = leg
                out[key] = {"dtype": "f32" if prec == "fp32" else "bf16x3", "value": round(leg["value"], 1), "ms": round(leg["ms_per_step"], 2),
                            "frac": round(leg["roofline"]["frac"], 4), "max_abs_vs_oracle": leg["max_abs_vs_oracle"], "tolerance": 1e-4}
                if "mfma_pipe_frac" in leg["roofline"]:
                    out[key]["pipe_frac"] = round(leg["roofline"]["mfma_pipe_frac"], 4)
            fx = bench_fx_chain(dev)
            details["fx_chain"] = fx
            out["fx_chain"] = {"value": round(fx["value"]), "ms": round(fx["ms_per_chain"], 4), "frac": round(fx["roofline"]["frac"], 4),
                               "max_abs_vs_oracle": fx["max_abs_vs_oracle"], "traffic": fx["roofline"]["traffic"],
                               "traffic_over_algorithmic": fx["roofline"]["traffic_over_algorithmic"],
                               "frac_on_traffic": fx["roofline"]["frac_on_traffic"],
                               "cpu_baseline": {"value": round(fx["cpu_baseline"]["value"], 2), "unit": "segments/s", "cores": 1, "kind": "port"},
                               "conv_reverb": {"value": round(fx["conv_reverb"]["value"]), "ms": round(fx["conv_reverb"]["ms_per_batch"], 3),
                                               "rel_dev_vs_oracle": fx["conv_reverb"]["rel_dev_vs_oracle"]}}
            nz = bench_input_normalizer()
            details["input_normalizer"] = nz
            out["input_normalizer"] = {"value": round(nz["value"], 1), "unit": nz["unit"]}
            out["file_to_file"] = {"value": round(f2f["value"], 4), "unit": f2f.get("unit", "s/song")}
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(enc_cfg, enc_sd, tcn_sd)
            xs, y_ref = cb.pop("x"), cb.pop("y_ref")
            # the headline precision against the oracle AT the headline's segment length: the segment the CPU baseline just converted
            # (reference role = input role, like its sample), one item through the same engine
            y_dev, _ = engine.step(xs.to(dev), xs.to(dev))
            out["max_abs_vs_oracle"] = float((y_dev.cpu() - y_ref).abs().max())
            out["tolerance"] = {"bf16": 1e-2, "fp32": 1e-4, "bf16x3": 1e-4}[args.precision]
            details["headline"]["max_abs_vs_oracle"] = {"value": out["max_abs_vs_oracle"], "tolerance": out["tolerance"],
                                                        "probe": f"1 reference + 1 input segment of 2x{SEG_LEN} (the CPU baseline's sample) vs oracle/networks_ref.py"}
            details["cpu_baseline"] = cb
            out["cpu_baseline"] = {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                                   "sample": f"oracle/networks_ref.py (torch-CPU fp32), 1 segment of 2x{SEG_LEN}, warm median of 3 at {cb['cores']} threads"}
        try:
            os.makedirs(os.path.dirname(DETAILS_PATH), exist_ok=True)
            with open(DETAILS_PATH, "w") as f:
                json.dump(dict(out, details=details), f, indent=1)
            out["details"] = os.path.relpath(DETAILS_PATH, REPO)
        except OSError:
            out["details"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

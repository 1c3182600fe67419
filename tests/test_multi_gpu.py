"""Self-verifying first visit to a multi-GPU box (reference loop being sharded: inference/style_transfer.py:144-162).
These tests SKIP on a one-GPU box (the pool this repository is developed on): the RCCL path has run with one rank only so far.
On a box with >= 2 GPUs they start one process per GPU over "nccl" and check that (1) N ranks really took part, (2) the sharded result is
bit-equal to one GPU's, (3) bench.py's N > 1 line carries the evidence (ranks_seen, per-rank segment counts, RCCL version, T1 / (N TN))."""
import json
import os
import subprocess
import sys

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _n_gpus():
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "MST_BENCH_SHARE_GPU",
                                                            "MST_DIST_BACKEND")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="8")
    return env


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


@pytest.mark.gpu
@pytest.mark.skipif(_n_gpus() < 2, reason="needs >= 2 GPUs (RCCL with more than one rank)")
@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_nccl_sharded_stem_is_bit_equal_to_one_gpu(precision):
    n = min(_n_gpus(), 8)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(REPO, "tests", "nccl_worker.py"), precision],
                       env=_clean_env(), capture_output=True, text=True, timeout=1200, cwd=REPO)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    print(out)
    assert out["world"] == n and out["devices"] >= n
    assert [tuple(v) for v in out["ranges"]] == [tuple(v) for v in out["ranges_expected"]]
    assert out["finite"] and out["bit_equal"], out


@pytest.mark.gpu
@pytest.mark.skipif(_n_gpus() < 2, reason="needs >= 2 GPUs (RCCL with more than one rank)")
def test_bench_two_gpus_over_rccl():
    """`python bench.py --gpus 2 --workload track60` exactly as the driver types it (bench.py starts its own ranks): one rank per GPU, nccl."""
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "track60"],
                       env=_clean_env(), capture_output=True, text=True, timeout=1500, cwd=REPO)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    print(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["scaling"] == "strong"
    assert out["comm"]["backend"] == "nccl" and out["comm"]["rccl_version"]
    t = out["track60"]
    assert t["segments"] == 1212 and t["segments_rank"] == [606, 606]
    assert "efficiency_t1_over_n_tn" in t and abs(t["efficiency_t1_over_n_tn"] - t["t1_ms_same_job"] / (2 * t["t_ms"])) < 1e-9
    assert t["efficiency_t1_over_n_tn"] > 0.5, t          # two real GPUs: well above what two ranks sharing one GPU can reach


@pytest.mark.gpu
def test_nccl_worker_script_with_one_rank():
    """The worker script of the tests above on whatever box this is, ONE rank over "nccl" (RCCL initialises, the engine's collectives run at
    world 1): keeps the script itself exercised where only one GPU exists."""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(REPO, "tests", "nccl_worker.py"), "bf16"],
                       env=_clean_env(), capture_output=True, text=True, timeout=900, cwd=REPO)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["world"] == 1 and out["bit_equal"] and out["finite"] and out["rccl_version"]

"""-m gpu: bench.py's N > 1 path end to end on a one-GPU box -- `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ...`
exactly as the driver launches it, with the two ranks sharing cuda:0 over gloo (RULGNN_BENCH_SHARE_GPU=1, a test hook: the numbers are not a
measurement).  Checks what no single-process test reaches: the rendezvous from the environment, the per-rank batches, barriers and the max
over ranks around the timed region, the strong-scaling leg, the data-parallel step with the real kernels, ONE JSON line from rank 0."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("extra", [[], ["--sync-bn"]])
def test_bench_two_ranks_prints_one_contract_line(extra):
    env = dict(os.environ, RULGNN_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--batch", "4096", "--reps", "2", "--no-roofline", "--no-cpu-baseline", "--no-families", "--no-rmse"] + extra
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["config"]["per_gpu_batch"] == 4096 and d["config"]["global_batch"] == 8192
    assert d["config"]["parallelism"] == ("dp2+syncbn" if extra else "dp2")
    assert d["value"] > 0 and abs(d["value"] - 8192 * 3 / (d["ms_per_step"] * 3e-3)) < 0.02 * d["value"]
    assert d["strong_scaling"]["global_batch"] == 4096 and d["strong_scaling"]["per_gpu_batch"] == 2048
    assert d["final_loss"] == d["final_loss"]            # not NaN
    assert d["vs_baseline"] is None and d["data"] == "synthetic"

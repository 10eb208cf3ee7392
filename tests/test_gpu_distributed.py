"""N > 1 path on the GPU box: two ranks (sharing the one GPU of the test box) replicate the server
key as a device blob and bootstrap their own shards. RCCL itself needs two GPUs and is exercised by
bench.py --gpus N; here the process group runs over gloo, and a one-rank world runs bench.py over RCCL."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("pname", ["PARAM_MESSAGE_2_CARRY_2", "PARAM_MESSAGE_2_CARRY_3"])
def test_two_ranks_replicated_key_sharded_batch(pname):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    port = "29611" if pname.endswith("2") else "29612"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.join(ROOT, "tests", "dist_gpu_worker.py"), ROOT, pname]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    assert res.stdout.count(" ok variant") == 2


def test_bench_over_rccl_with_one_rank():
    """bench.py's multi-rank code path with the process group on RCCL (backend "nccl", device
    tensors in the barrier and the reductions, the key blob broadcast from device memory), on the
    one GPU of the test box: a world of one rank. The line must say that RCCL carried it."""
    import json

    env = dict(os.environ, CAMD_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29613", RANK="0",
               WORLD_SIZE="1", LOCAL_RANK="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--batch", "2048",
           "--no-cpu-baseline", "--no-host-buffer-pass"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    line = json.loads(res.stdout.strip().splitlines()[-1])
    assert line["backend"] == "nccl" and line["ranks_reporting"] == 1 and line["n_gpus"] == 1
    assert line["key_broadcast_ms"] is not None and line["value"] > 0
    assert line["decrypt_differs_from_cleartext"] <= 4

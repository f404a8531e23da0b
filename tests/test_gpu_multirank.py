"""The N > 1 launch path on a 1-GPU box: two ranks started exactly like the driver starts bench.py
(python -m torch.distributed.run --nproc-per-node 2 ...), both on cuda:0 (BENCH_ONE_DEVICE=1).
RCCL is tried first; a 1-GPU box may refuse two ranks on one device, in which case the same code runs
over gloo (the exchange tensors then travel through host memory).  Checks the JSON contract, the
sharded exchange (scatter -> decode -> gather) and that the result does not depend on the number of
ranks (tools/config4_recorded.py: recorded frames, checksum of the gathered clouds)."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _torchrun(script_args, nproc, backend):
    env = dict(os.environ, BENCH_ONE_DEVICE="1", BENCH_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + script_args
    return subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)


def _last_json(stdout):
    for line in reversed(stdout.strip().splitlines()):
        try:
            return json.loads(line)
        except ValueError:
            continue
    raise AssertionError("no JSON line in: " + stdout[-2000:])


def _two_ranks(script_args):
    p = _torchrun(script_args, 2, "nccl")
    backend = "nccl"
    if p.returncode != 0:      # two ranks on one GPU: RCCL may refuse ("Duplicate GPU detected")
        p = _torchrun(script_args, 2, "gloo")
        backend = "gloo"
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    return _last_json(p.stdout), backend


def test_bench_two_ranks_with_exchange():
    line, backend = _two_ranks(["bench.py", "--gpus", "2", "--steps", "5", "--warmup", "2", "--frames", "32",
                                "--no-cpu", "--exchange"])
    assert line["n_gpus"] == 2 and line["steps"] == 5 and line["value"] > 0
    assert line["config"]["sharding"].startswith("frames x2")
    ex = line["exchange"]
    assert ex["frames_total"] == 64 and ex["scatter_packets_ms"] > 0 and ex["gather_xyz_ms"] > 0
    print("two-rank transport:", backend, ex)


def test_config4_recorded_frames_same_result_for_1_and_2_ranks():
    one = subprocess.run([sys.executable, "tools/config4_recorded.py", "--frames", "64", "--reps", "1"],
                         capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert one.returncode == 0, one.stderr[-3000:]
    a = _last_json(one.stdout)
    b, backend = _two_ranks(["tools/config4_recorded.py", "--frames", "64", "--reps", "1"])
    assert a["n_gpus"] == 1 and b["n_gpus"] == 2
    assert a["recorded_frames"] == b["recorded_frames"] > 0
    assert a["xyz_checksum"] == b["xyz_checksum"] and a["nonzero_points"] == b["nonzero_points"] > 0
    print("config 4:", backend, b)

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


def _n_gpus():
    import torch
    return torch.cuda.device_count()


def _torchrun(script_args, nproc, backend, one_device=True):
    env = dict(os.environ, BENCH_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if one_device:
        env["BENCH_ONE_DEVICE"] = "1"
    else:
        env.pop("BENCH_ONE_DEVICE", None)
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
    """Two ranks.  With two or more GPUs visible: one rank per GPU over RCCL, and a failure IS a failure (no fallback).
    On a one-GPU box RCCL refuses two ranks on one device ("Duplicate GPU detected"); only then the same code runs over
    gloo with both ranks on cuda:0, and the returned backend says so."""
    if _n_gpus() >= 2:
        p = _torchrun(script_args, 2, "nccl", one_device=False)
        assert p.returncode == 0, "RCCL run on %d GPUs failed:\n" % _n_gpus() + p.stdout[-2000:] + p.stderr[-3000:]
        return _last_json(p.stdout), "nccl"
    p = _torchrun(script_args, 2, "nccl")
    backend = "nccl"
    if p.returncode != 0:
        p = _torchrun(script_args, 2, "gloo")
        backend = "gloo"
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    return _last_json(p.stdout), backend


def test_plain_bench_gpus_2_starts_two_ranks_itself():
    """`python bench.py --gpus 2 ...` with NO launcher in the test: bench.py re-executes itself through
    torch.distributed.run (bench.plan_launch) and rank 0's line says n_gpus == 2.  Two GPUs visible: RCCL, one rank per
    GPU, no fallback.  One GPU: both ranks on cuda:0 (BENCH_ONE_DEVICE), RCCL first, gloo when it refuses the duplicate."""
    args = [sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--frames", "32", "--no-cpu",
            "--placement", "first", "--no-loss-paths", "--no-extras"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BENCH_ONE_DEVICE")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    if _n_gpus() >= 2:
        tries = [dict(env, BENCH_BACKEND="nccl")]
    else:
        tries = [dict(env, BENCH_BACKEND="nccl", BENCH_ONE_DEVICE="1"), dict(env, BENCH_BACKEND="gloo", BENCH_ONE_DEVICE="1")]
    for e in tries:
        p = subprocess.run(args, capture_output=True, text=True, env=e, timeout=900, cwd=ROOT)
        if p.returncode == 0:
            break
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    line = _last_json(p.stdout)
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["value"] > 0, line
    assert line["config"]["sharding"].startswith("frames x2")
    if e["BENCH_BACKEND"] == "nccl":
        assert line["rccl_ranks"] == 2
    # a box with fewer GPUs than asked for is an error, never a smaller run
    if _n_gpus() < 4:
        p = subprocess.run([sys.executable, "bench.py", "--gpus", "4", "--steps", "1"], capture_output=True, text=True,
                           env=env, timeout=300, cwd=ROOT)
        assert p.returncode != 0 and "GPU(s) visible" in p.stderr


def test_bench_two_ranks_with_exchange():
    line, backend = _two_ranks(["bench.py", "--gpus", "2", "--steps", "5", "--warmup", "2", "--frames", "32",
                                "--no-cpu", "--exchange", "--placement", "first", "--no-loss-paths"])
    assert line["n_gpus"] == 2 and line["steps"] == 5 and line["value"] > 0
    assert line["config"]["sharding"].startswith("frames x2")
    ex = line["exchange"]
    assert ex["frames_total"] == 64 and ex["scatter_packets_ms"] > 0 and ex["gather_xyz_ms"] > 0
    if backend == "nccl":
        assert line["rccl_ranks"] == 2 and "RCCL" in line["collective_backend"], line
    else:
        assert line["rccl_ranks"] is None and "RCCL was not used" in line["collective_backend"], line
    print("two-rank transport:", backend, line["collective_backend"], ex)


@pytest.mark.parametrize("workload", ["batch512", "fused4"])
def test_bench_two_ranks_other_workloads_with_exchange(workload):
    """configs[3] (a fixed 512-frame batch split over the ranks) and configs[4] (four sensors per tick) through the same
    two-rank launch + exchange, so that the driver's SCALE run needs no code that has not run before."""
    line, backend = _two_ranks(["bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--frames", "32", "--workload",
                                workload, "--no-cpu", "--exchange", "--placement", "first", "--no-loss-paths"])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["exchange"]["gather_xyz_ms"] > 0
    assert line["scaling"] == ("strong" if workload == "batch512" else "weak")
    assert (line["rccl_ranks"] == 2) == (backend == "nccl")


def test_batch512_exchange_checksum_does_not_depend_on_the_number_of_ranks():
    """`bench.py --gpus 2 --workload batch512 --exchange` -- configs[3]'s 512 frames scattered from rank 0,
    decoded by two ranks, gathered back -- must leave the clouds one rank leaves (exchange.xyz_checksum).  With two or more
    GPUs visible the two ranks run over RCCL, one per GPU, and anything else is a failure (no skip, no gloo)."""
    one = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--workload", "batch512",
                          "--no-cpu", "--exchange", "--placement", "first", "--no-loss-paths", "--no-extras"],
                         capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert one.returncode == 0, one.stderr[-3000:]
    a = _last_json(one.stdout)
    b, backend = _two_ranks(["bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "batch512", "--no-cpu",
                             "--exchange", "--placement", "first", "--no-loss-paths", "--no-extras"])
    if _n_gpus() >= 2:
        assert backend == "nccl" and b["rccl_ranks"] == 2, b
    assert a["exchange"]["frames_total"] == b["exchange"]["frames_total"] == 512
    assert a["exchange"]["xyz_checksum"] == b["exchange"]["xyz_checksum"] != 0, (a["exchange"], b["exchange"])


def test_eight_ranks_dry_run_batch512_exchange_on_whatever_is_here():
    """the launch the driver's SCALE run makes on an 8-GPU node -- `bench.py --gpus 8 --workload batch512
    --exchange`, eight ranks started by bench.py itself -- before there is such a node: with eight or more GPUs visible over RCCL,
    one rank per GPU (no fallback); otherwise all eight ranks on cuda:0 over gloo (BENCH_ONE_DEVICE=1: same code, the exchange
    tensors travel through host memory).  The 512 frames scattered from rank 0, decoded by eight ranks and gathered back must
    leave the clouds ONE rank leaves; the line carries every rank's own timing and kernel."""
    base = ["--steps", "2", "--warmup", "1", "--workload", "batch512", "--no-cpu", "--exchange", "--no-loss-paths", "--no-extras"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BENCH_ONE_DEVICE", "BENCH_BACKEND")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    one = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--placement", "first"] + base, capture_output=True, text=True,
                         cwd=ROOT, timeout=900, env=env)
    assert one.returncode == 0, one.stderr[-3000:]
    a = _last_json(one.stdout)
    if _n_gpus() >= 8:
        env8 = dict(env, BENCH_BACKEND="nccl")
    else:
        env8 = dict(env, BENCH_BACKEND="gloo", BENCH_ONE_DEVICE="1")
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "8"] + base, capture_output=True, text=True, cwd=ROOT, timeout=1500, env=env8)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    b = _last_json(p.stdout)
    assert b["n_gpus"] == 8 and b["validated"] is True and b["value"] > 0, b
    assert b["scaling"] == "strong" and b["config"]["sharding"].startswith("frames x8")
    assert a["exchange"]["frames_total"] == b["exchange"]["frames_total"] == 512
    assert a["exchange"]["xyz_checksum"] == b["exchange"]["xyz_checksum"] != 0, (a["exchange"], b["exchange"])
    pr = b["per_rank"]
    assert sum(pr["kernel"].values()) == 8 and 0 < pr["ms_min"] <= pr["ms_max"] and 0 <= pr["slowest_rank"] < 8, pr
    assert abs(b["ms_per_step"] - pr["ms_max"]) <= 0.25 * pr["ms_max"] + 0.05, (b["ms_per_step"], pr)   # the line is the slowest rank's time
    if _n_gpus() >= 8:
        assert b["rccl_ranks"] == 8
    else:
        # eight ranks share one device: the placement search (two more output sets per rank) stands down where memory is short,
        # and says so; with 288 GB it normally runs
        assert all(x in ("refine", "first:skipped") for x in pr["placement"]), pr
    print("eight ranks:", b["collective_backend"], pr, b["exchange"])


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


def test_rccl_backend_runs_the_p2p_group_on_this_box():
    """RCCL itself (backend "nccl") on the one GPU that is here: a world-size-1 process group and a
    grouped self send / recv through the same `parallel._p2p` helper scatter_frames / gather_frames use
    (two ranks cannot share a GPU under RCCL, so the 2-rank tests above fall back to gloo)."""
    code = r'''
import os, torch, torch.distributed as dist
from ouster_sdk_amd import parallel
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ["PORT"], RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
a = torch.arange(1 << 20, dtype=torch.int32, device="cuda")
b = torch.zeros_like(a)
parallel._p2p([dist.P2POp(dist.isend, a, 0), dist.P2POp(dist.irecv, b, 0)])
torch.cuda.synchronize()
assert torch.equal(a, b)
mine = parallel.scatter_frames(a.view(64, -1), 64, (a.numel() // 64,), torch.int32, "cuda")
out = parallel.gather_frames(mine, 64)
assert torch.equal(out.view(-1), a)
print("rccl ok", parallel.max_over_ranks(1.5, torch.device("cuda")))
dist.destroy_process_group()
'''
    env = dict(os.environ, PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert p.returncode == 0 and "rccl ok 1.5" in p.stdout, p.stdout[-1000:] + p.stderr[-3000:]

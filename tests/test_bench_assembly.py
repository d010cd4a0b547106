"""bench.py's multi-rank bookkeeping without a GPU or a real process group: the first
`torchrun --nproc-per-node 8 bench.py --gpus 8` must not fail on result assembly.  Driver contract: EXACTLY K timed steps
between barrier + synchronise fences, elapsed = MAX over ranks, value = whole-job images / that time, n_gpus = N,
scaling weak, one JSON line."""
import json
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class StubDist:
    """What bench.timed_steps uses of torch.distributed: barrier() and all_reduce(t, op=MAX); the stub plays the slowest of
    the other ranks."""

    class ReduceOp:
        MAX = "max"

    def __init__(self, slowest_other_rank_s):
        self.other = slowest_other_rank_s
        self.events = []

    def barrier(self):
        self.events.append("barrier")

    def all_reduce(self, t, op=None):
        assert op == self.ReduceOp.MAX and t.dtype == torch.float64 and t.numel() == 1
        self.events.append("all_reduce")
        t[0] = max(float(t[0]), self.other)


def _args(**kw):
    import bench
    argv, sys.argv = sys.argv, ["bench.py"] + [str(v) for kv in kw.items() for v in ("--" + kv[0].replace("_", "-"), kv[1])]
    try:
        return bench.parse()
    finally:
        sys.argv = argv


def test_timed_region_and_headline_at_world_8():
    import bench
    a = _args(gpus=8, steps=7, warmup=2)
    calls, syncs = [], []
    dist = StubDist(slowest_other_rank_s=0.25)

    def step(x, y):
        calls.append((x, y))
        time.sleep(0.001)
        return ("logits", torch.tensor(1.5), torch.tensor(0.25))
    elapsed, out = bench.timed_steps(step, "x", "y", a.steps, 8, dist, torch.device("cpu"), sync=lambda: syncs.append(1))
    assert len(calls) == 7 and out[0] == "logits"                    # exactly K steps
    assert dist.events == ["barrier", "barrier", "all_reduce"]      # a fence on both sides, then the MAX over ranks
    assert len(syncs) == 4                                          # synchronise around each barrier
    assert elapsed == 0.25                                          # the slowest rank's time, not this rank's ~7 ms
    res = bench.headline(a, elapsed, 128, 8, False, 1.5, 0.25)
    assert res["n_gpus"] == 8 and res["steps"] == 7 and res["warmup"] == 2 and res["scaling"] == "weak"
    assert res["value"] == 7 * 128 * 8 / 0.25 and res["unit"] == "images/sec" and res["higher_is_better"] is True
    assert abs(res["ms_per_step"] - 0.25 / 7 * 1e3) < 1e-9
    assert res["config"]["global_batch"] == 1024 and res["config"]["parallelism"] == "dp8"
    assert res["metric"].startswith("images/sec (train step, CDF+ADMM) ResNet-20 8-bit") and res["vs_baseline"] is None
    assert res["dtype"] == "f32" and res["data"] == "synthetic" and "model" not in res["config"]
    json.dumps(res)                                                 # one JSON line: everything serialisable


def test_single_rank_takes_no_collective():
    import bench
    a = _args(steps=3, warmup=0)
    dist = StubDist(1e9)
    elapsed, _ = bench.timed_steps(lambda x, y: (None, torch.tensor(0.0), None), None, None, a.steps, 1, dist, torch.device("cpu"),
                                   sync=lambda: None)
    assert dist.events == [] and elapsed < 1.0
    res = bench.headline(a, elapsed, 128, 1, False, 0.0, None)
    assert res["n_gpus"] == 1 and res["config"]["parallelism"] == "dp1" and res["config"]["final_trans_loss"] is None


def test_office_headline_counts_both_passes():
    import bench
    a = _args(model="resnet50_dann", batch=28, steps=5, gpus=8)
    res = bench.headline(a, 1.0, 2 * 28, 8, True, 3.0, 1.0)
    assert res["value"] == 5 * 56 * 8 and "Office-31" in res["config"]["workload"] and res["config"]["global_batch"] == 224


# ---- round 5: `python bench.py --gpus N` must start N ranks by itself, and a rank count that differs
#      from --gpus must stop every rank with a non-zero exit instead of printing n_gpus from somewhere else
def test_launcher_command_line_for_8_gpus():
    import bench
    argv = ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    a = _args(gpus=8, steps=20, warmup=5)
    cmd = bench.launcher_command(a, argv, port=29999)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29999"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == argv                                      # the ranks get the same arguments, --gpus 8 included


def test_self_launch_is_a_child_process_and_relays_its_exit_code(monkeypatch):
    import subprocess
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(os, "execv", lambda *a, **k: (_ for _ in ()).throw(AssertionError("never exec from bench.py")))
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    try:
        bench.main()
        raise AssertionError("main() must exit with the child's code")
    except SystemExit as e:
        assert e.code == 7
    assert "--nproc-per-node=4" in seen["cmd"] and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_world_size_mismatch_exits_non_zero():
    import subprocess
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "--gpus 8 but WORLD_SIZE=2" in r.stderr and r.stdout.strip() == ""


def test_gpus_2_dry_run_through_the_self_launch():
    """The whole chain on the CPU: parent -> torch.distributed.run child -> 2 gloo ranks -> fences, MAX over ranks -> one JSON
    line on the parent's stdout with n_gpus 2."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["MASTER_PORT"] = str(port)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1", "--dry-run"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 5 and res["config"]["parallelism"] == "dp2" and res["metric"].startswith("dry run")
    assert res["ms_per_step"] >= 2.0                                # rank 1 sleeps 2 ms per step: the slowest rank's clock
    assert res["value"] == 5 * 128 * 2 / (res["ms_per_step"] * 5e-3) or abs(res["value"] * res["ms_per_step"] * 5e-3 - 1280) < 1e-6

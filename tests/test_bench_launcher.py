"""`bench.py --gpus N`: the supervisors that start, watch and -- when a halo transport fails the way RCCL fails (one rank
exits, or one rank never arrives and the others hang) -- replace FRESH worker processes with the next transport of the
chain.  No GPU here: BFLBM_BENCH_FAKE_WORKER=1 swaps the measurement for a rendezvous check of the workers (a gloo
all-reduce on the fresh port), everything else -- torch.distributed.run, the supervisors' gloo group, the process
handling, the line that is printed -- is the code the driver's N > 1 command runs.
The measured path itself is rehearsed on the GPU box (tests/test_gpu_slabs.py)."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, args=(), by_hand=False, world=2):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, BFLBM_BENCH_FAKE_WORKER="1", **extra_env)
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1", *args]
    if by_hand:
        cmd = [sys.executable] + tail
    else:                                                 # the driver's command
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
               "--master-addr", "127.0.0.1", "--master-port", str(port)] + tail
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    return out, lines


def _no_leftover_workers():
    ps = subprocess.run(["ps", "-eo", "pid,cmd"], capture_output=True, text=True).stdout
    mine = [l for l in ps.splitlines() if "bench.py" in l and "--attempt-timeout 6.5" in l]
    assert not mine, mine


def test_first_transport_completes_and_the_next_family_is_timed_beside_it():
    out, lines = _run({}, args=("--second-transport",))
    assert out.returncode == 0, out.stderr[-2000:]
    assert len(lines) == 1, out.stdout                     # ONE line on stdout, whatever the workers printed
    r = json.loads(lines[0])
    tried = r["config"]["launcher"]["transports_tried"]
    assert [t["transport"] for t in tried] == ["rccl", "peer-copy"] and all(t["ok"] for t in tried)
    assert tried[1].get("informational") is True
    assert r["config"]["halo_transport"].startswith("rccl, staged")
    assert r["config"]["second_transport"]["halo_transport"].startswith("peer")
    assert r["n_gpus"] == 2


def test_by_default_only_the_first_transport_that_completes_runs():
    out, lines = _run({})
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads(lines[0])
    assert [t["transport"] for t in r["config"]["launcher"]["transports_tried"]] == ["rccl"]
    assert "second_transport" not in r["config"]


def test_a_failing_and_a_hanging_transport_are_replaced_by_fresh_workers():
    """rccl: the last rank exits with an error before the rendezvous (its peer would wait for it for ever);
    peer-copy: the single worker never finishes (killed at the limit); peer-kernel completes and produces the line."""
    out, lines = _run({"BFLBM_BENCH_FAIL": "rccl:exit,peer-copy:hang"}, args=("--attempt-timeout", "6.5", "--no-second-transport"), by_hand=True)
    assert out.returncode == 0, out.stderr[-2000:]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    tried = r["config"]["launcher"]["transports_tried"]
    assert [(t["transport"], t["ok"]) for t in tried] == [("rccl", False), ("peer-copy", False), ("peer-kernel", True)]
    assert "exited with an error" in tried[0]["note"] and "limit" in tried[1]["note"]
    assert r["config"]["halo_transport"].startswith("peer, one process drives all GPUs; a gather kernel")
    assert "FAILED" in out.stderr
    _no_leftover_workers()


def test_no_transport_completes_is_an_error_not_a_line():
    out, lines = _run({"BFLBM_BENCH_FAIL": "rccl:exit,rccl-direct:hang"}, args=("--transport", "rccl,rccl-direct", "--attempt-timeout", "6.5"), world=3)
    assert out.returncode != 0
    assert not lines, out.stdout
    assert "no transport completed" in out.stderr
    _no_leftover_workers()

"""bench.py's own N-worker launcher (`python bench.py --gpus N` without torch.distributed.run) on CPU, with stub workers:
a worker that dies must end the whole job within seconds (its peers would otherwise sit in an RCCL collective until the
NCCL timeout, longer than the driver's limit - VERDICT r2 weak #4), the in-stream -> async fallback starts a FRESH set of
workers exactly once and only after an ordinary error status, and a hung job ends at the launcher's deadline."""
import argparse
import json
import os
import sys
import textwrap
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _stub(tmp_path, body):
    p = tmp_path / "stub_worker.py"
    p.write_text("import os, sys, time, json\nrank = int(os.environ['RANK'])\nworld = int(os.environ['WORLD_SIZE'])\n"
                 + textwrap.dedent(body))
    return [sys.executable, str(p)]


def _args(gpus, dtype="fp32"):
    return argparse.Namespace(gpus=gpus, dtype=dtype)


def test_dead_worker_ends_the_job_at_once(tmp_path, capfd, monkeypatch):
    import bench
    monkeypatch.delenv("UMPR_COMM_ASYNC", raising=False)
    cmd = _stub(tmp_path, """
        if rank == 1:
            sys.stderr.write("rank 1: simulated failure during init\\n")
            sys.exit(1)
        time.sleep(120)      # a peer blocked in a collective
        """)
    t0 = time.time()
    with pytest.raises(SystemExit) as e:
        bench.spawn_workers(_args(3), argv=[], cmd=cmd, deadline_s=100)
    dt = time.time() - t0
    assert e.value.code == 1 and dt < 15, dt
    err = capfd.readouterr().err
    assert "rank 1 exited with status 1" in err and "simulated failure during init" in err


def test_successful_job_relays_rank0_line(tmp_path, capfd):
    import bench
    cmd = _stub(tmp_path, """
        assert os.environ["MASTER_ADDR"] == "127.0.0.1" and os.environ["LOCAL_RANK"] == str(rank)
        if rank == 0:
            sys.stderr.write("[bench] progress note\\n")
            print(json.dumps({"value": 1.0, "n_gpus": world}))
        """)
    bench.spawn_workers(_args(2), argv=[], cmd=cmd, deadline_s=60)
    out, err = capfd.readouterr()
    assert json.loads(out.strip()) == {"value": 1.0, "n_gpus": 2}
    assert "progress note" in err


def test_in_stream_failure_falls_back_to_async_in_fresh_workers(tmp_path, capfd, monkeypatch):
    import bench
    monkeypatch.delenv("UMPR_COMM_ASYNC", raising=False)
    marker = tmp_path / "attempts"
    cmd = _stub(tmp_path, f"""
        open({str(marker)!r} + str(rank), "a").write(os.environ.get("UMPR_COMM_ASYNC", "-") + "\\n")
        if os.environ.get("UMPR_COMM_ASYNC") != "1":
            if rank == 1:
                sys.stderr.write("RuntimeError: in-stream exchange refused\\n")
                sys.exit(3)
            time.sleep(120)
        if rank == 0:
            print(json.dumps({{"value": 2.0, "comm_fallback": os.environ.get("UMPR_BENCH_FALLBACK")}}))
        """)
    t0 = time.time()
    bench.spawn_workers(_args(2, dtype="bf16"), argv=[], cmd=cmd, deadline_s=100)      # bf16: in-stream is the default form
    assert time.time() - t0 < 20
    out, err = capfd.readouterr()
    line = json.loads(out.strip())
    assert line["value"] == 2.0 and "rank 1 exited 3" in line["comm_fallback"]
    assert open(str(marker) + "0").read().split() == ["-", "1"]          # two process sets, the second with the switch
    assert "fresh set of workers" in err


def test_no_fallback_when_already_async_or_killed_by_signal(tmp_path, capfd, monkeypatch):
    import bench
    monkeypatch.delenv("UMPR_COMM_ASYNC", raising=False)
    marker = tmp_path / "n"
    # fp32: the async form is already the default - nothing to fall back to
    cmd = _stub(tmp_path, f"""
        open({str(marker)!r}, "a").write("x")
        sys.exit(2)
        """)
    with pytest.raises(SystemExit):
        bench.spawn_workers(_args(1, dtype="fp32"), argv=[], cmd=cmd, deadline_s=60)
    assert open(marker).read() == "x"
    # bf16 but the worker died from a signal (what a GPU fault looks like): never retried
    marker.write_text("")
    cmd = _stub(tmp_path, f"""
        import signal
        open({str(marker)!r}, "a").write("y")
        os.kill(os.getpid(), signal.SIGABRT)
        """)
    with pytest.raises(SystemExit):
        bench.spawn_workers(_args(1, dtype="bf16"), argv=[], cmd=cmd, deadline_s=60)
    assert open(marker).read() == "y"


def test_deadline_kills_a_hung_job(tmp_path, capfd):
    import bench
    cmd = _stub(tmp_path, "time.sleep(120)\n")
    t0 = time.time()
    with pytest.raises(SystemExit):
        bench.spawn_workers(_args(2, dtype="bf16"), argv=[], cmd=cmd, deadline_s=2)
    assert time.time() - t0 < 20
    assert "hit the deadline" in capfd.readouterr().err

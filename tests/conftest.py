import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # torch's own DataLoader pin_memory helper calls a deprecated overload once per tensor
    config.addinivalue_line("filterwarnings", "ignore:The argument 'device' of Tensor:DeprecationWarning")


# ---- two-rank data-parallel check on real kernels (tools/check_dp_gpu.py) ---------------------------------------
# A process that has initialised the GPU must not fork+exec another GPU program on this pool, so the two-rank job is
# started here, at session start, BEFORE anything in this process touches the GPU; tests/test_gpu_e2e.py collects it.
DP_CHECK = {"proc": None, "log": os.path.join(ROOT, "gpurun_out", "dp_check.log")}
# Same rule for the jobs of tools/run_gpu_children.py, which run one after the other in ONE child: six test_conv3x3 cases
# with UMPR_WINO_F4=0 and =1 (read when the library loads; default 2 = F(4x4,3x3) in forward and backward) and the world-1 RCCL run of
# the gradient exchange.  Collected by test_conv3x3_winograd_modes / test_gradient_exchange_on_rccl_at_world_one.
CHILDREN = {"proc": None, "rc": os.path.join(ROOT, "gpurun_out", "gpu_children.rc")}


def child_result(name, timeout=900):
    """(exit code, log text) of one job of tools/run_gpu_children.py; waits for the launcher."""
    p = CHILDREN["proc"]
    assert p is not None, "the child jobs were not started (no /dev/kfd, or GPU tests deselected)"
    p.wait(timeout=timeout)
    rcs = dict(line.split() for line in open(CHILDREN["rc"]).read().splitlines() if line.strip())
    assert name in rcs, (name, rcs)
    return int(rcs[name]), open(os.path.join(ROOT, "gpurun_out", name + ".log")).read()


def pytest_sessionstart(session):
    import subprocess
    mexpr = session.config.getoption("-m") or ""
    if "not gpu" in mexpr or not os.path.exists("/dev/kfd"):   # a GPU box and GPU tests not deselected
        return
    if session.config.getoption("collectonly", False) or os.environ.get("UMPR_TEST_CHILD"):
        return
    os.makedirs(os.path.dirname(DP_CHECK["log"]), exist_ok=True)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["OMP_NUM_THREADS"] = "4"
    with open(DP_CHECK["log"], "w") as f:
        DP_CHECK["proc"] = subprocess.Popen(
            [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
             "127.0.0.1", "--master-port", "29531", os.path.join(ROOT, "tools", "check_dp_gpu.py")],
            stdout=f, stderr=subprocess.STDOUT, env=env, cwd=ROOT)
    CHILDREN["proc"] = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "run_gpu_children.py")], env=env, cwd=ROOT,
                                        stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def pytest_sessionfinish(session, exitstatus):
    for p in (DP_CHECK["proc"], CHILDREN["proc"]):
        if p is not None and p.poll() is None:
            p.kill()


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden


def has_gpu():
    return torch.cuda.is_available()

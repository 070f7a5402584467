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
# Same rule for the other settings of UMPR_WINO_F4 (read when the library loads; default 1 = F(4x4,3x3) in backward only):
# 0 = F(2x2,3x3) everywhere, 2 = F(4x4,3x3) in forward as well.  Each runs four test_conv3x3 cases in one child test run,
# collected by tests/test_gpu_parity.py::test_conv3x3_winograd_modes.
WINO_CHECKS = {m: {"proc": None, "log": os.path.join(ROOT, "gpurun_out", f"wino_f4_mode{m}_check.log")} for m in ("0", "2")}
WINO_CASES = "test_conv3x3 and (2-64-96-56 or 3-40-200-28 or 4-33-65-28 or 1-256-512-28)"


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
    for mode, c in WINO_CHECKS.items():
        with open(c["log"], "w") as f:
            c["proc"] = subprocess.Popen(
                [sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-q", "-x", "-m", "gpu", "-p",
                 "no:cacheprovider", "-k", WINO_CASES],
                stdout=f, stderr=subprocess.STDOUT, env=dict(env, UMPR_WINO_F4=mode, UMPR_TEST_CHILD="1"), cwd=ROOT)


def pytest_sessionfinish(session, exitstatus):
    for p in [DP_CHECK["proc"]] + [c["proc"] for c in WINO_CHECKS.values()]:
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

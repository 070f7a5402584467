#!/usr/bin/env python3
"""Started by tests/conftest.py at session start, BEFORE the test process touches the GPU (a process that has initialised
the GPU must not fork+exec another GPU program on this pool).  Runs, one after the other so that few processes share the card:
  * six test_conv3x3 cases with UMPR_WINO_F4=0 and again with =1 (the switch is read when the library loads; default 2),
  * tools/check_exchange_world1.py (gradient exchange on the RCCL backend at world size 1),
  * ten model-level oracle comparisons of the text path with UMPR_POISON_WS=1 (workspaces pre-filled with NaN bytes).
This launcher itself never touches the GPU.  Each job writes gpurun_out/<name>.log; the exit codes go to
gpurun_out/gpu_children.rc as `<name> <rc>` lines."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
WINO_CASES = "test_conv3x3 and (2-64-96-56 or 3-40-200-28 or 4-33-65-28 or 1-256-512-28 or 5-256-256-14 or 2-129-257-14)"


def main():
    os.makedirs(OUT, exist_ok=True)
    env = dict(os.environ, UMPR_TEST_CHILD="1")
    jobs = [(f"wino_f4_mode{m}_check", dict(env, UMPR_WINO_F4=m),
             [sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-q", "-x", "-m", "gpu", "-p",
              "no:cacheprovider", "-k", WINO_CASES]) for m in ("0", "1")]
    jobs.append(("exchange_world1_check", env, [sys.executable, os.path.join(ROOT, "tools", "check_exchange_world1.py")]))
    # UMPR_POISON_WS=1 (read when the library loads): the fused text path's entry points fill their workspace / arena with NaN
    # bytes first, so a read of memory the call did not write shows in these oracle comparisons
    jobs.append(("poison_ws_check", dict(env, UMPR_POISON_WS="1"),
                 [sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-q", "-x", "-m", "gpu", "-p",
                  "no:cacheprovider", "-k", "umpr_r_golden or umpr_r_small_batches or kernel_size_and_sentence or embedding_widths"]))
    with open(os.path.join(OUT, "gpu_children.rc"), "w") as rcf:
        for name, e, cmd in jobs:
            with open(os.path.join(OUT, name + ".log"), "w") as f:
                rc = subprocess.call(cmd, stdout=f, stderr=subprocess.STDOUT, env=e, cwd=ROOT, timeout=900)
            rcf.write(f"{name} {rc}\n")
            rcf.flush()


if __name__ == "__main__":
    main()

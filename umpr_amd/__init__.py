"""umpr_amd - MI355X-native UMPR hot path (host mirror of the reference API over libumpr_hip.so)."""
import os

# The step runs on up to six HIP streams at once (main, text path, weight gradients, RCCL's own, the optimiser's early slice,
# the H2D copy stream).  ROCm maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); streams that share a queue
# serialise - measured +6.5 ms per step in the RCCL rehearsal with 4 queues, +1.1 ms with 8.  Must be set before the HIP
# runtime initialises, i.e. before the first torch.cuda call of the process.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

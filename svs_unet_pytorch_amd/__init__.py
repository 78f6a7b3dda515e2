"""MI355X-native hot path of SVS-UNet-PyTorch: host code over libsvs_hip.so."""
import os as _os

# The backward pass runs the weight-gradient GEMMs on a second HIP stream (csrc/net.hip).  ROCm's runtime multiplexes
# all streams of a process onto GPU_MAX_HW_QUEUES hardware queues (default 4); once RCCL and torch have created
# theirs, the library's side stream shares a queue with the compute stream and the two serialise (measured: 4.5 ms per
# step instead of 4.0 when the process group is initialised first).  Eight queues keep it on a queue of its own.
# Must be set before the HIP runtime initialises, i.e. before the first torch.cuda call of the process; an explicit
# setting by the user wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

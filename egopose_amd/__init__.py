"""egopose_amd: MI355X-native PPO rollout+update hot path of EgoPose.

Only what the hot path needs lives here: ``csrc/`` (HIP kernels, the C-ABI, the host
physics boundary and the lockstep rollout engine), the ctypes binding, the batched
rollout driver, and ``compat/`` -- same-named mirrors of the reference's Python
interface for this path (see DESIGN.md / INTEGRATION.md).
"""
__version__ = "0.1.0"

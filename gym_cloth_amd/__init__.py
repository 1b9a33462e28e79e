"""gym_cloth_amd: MI355X-native drop-in for the physics hot path of DanielTakeshi/gym-cloth.

`gym_cloth/physics` (Point / Spring / Cloth.update / Gripper) is replaced by hand-written HIP kernels
for gfx950 behind a C ABI (include/clothhip.h, libclothhip.so), called through ctypes.  The host side
stays Python and keeps the reference's names: Cloth, Gripper, ClothEnv (+ the batched ClothBatch /
ClothVecEnv that the reference does not have).
"""
from ._lib import ClothHipError, F32, F64  # noqa: F401
from .batch import ClothBatch, make_schedules, schedule_bounds  # noqa: F401

__all__ = ["ClothBatch", "ClothHipError", "F32", "F64", "make_schedules", "schedule_bounds"]

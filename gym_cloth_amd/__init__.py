"""gym_cloth_amd: MI355X-native drop-in for the physics hot path of DanielTakeshi/gym-cloth.

`gym_cloth/physics` (Point / Spring / Cloth.update / Gripper) is replaced by hand-written HIP kernels
for gfx950 behind a C ABI (include/clothhip.h, libclothhip.so), called through ctypes.  The host side
stays Python and keeps the reference's names: Cloth, Gripper, ClothEnv (+ the batched ClothBatch /
ClothVecEnv that the reference does not have).
"""
from ._lib import ClothHipError, F32, F64  # noqa: F401
from .batch import ClothBatch, make_schedules, schedule_bounds  # noqa: F401

__all__ = ["ClothBatch", "ClothHipError", "F32", "F64", "make_schedules", "schedule_bounds", "register_gym_env"]


def register_gym_env(env_id='cloth-v0'):
    """gym_cloth/__init__.py:1-5 registers 'cloth-v0' -> gym_cloth.envs:ClothEnv on import. gym is not a dependency
    of the stepper, so the same registration happens only when gym is importable. Returns True if registered."""
    try:
        from gym.envs.registration import register
    except ImportError:
        return False
    try:
        register(id=env_id, entry_point='gym_cloth_amd.envs:ClothEnv')
    except Exception as e:                       # gym raises its own Error type when the id is already registered
        if 'register' not in str(e).lower() and 'exist' not in str(e).lower():
            raise
    return True


register_gym_env()

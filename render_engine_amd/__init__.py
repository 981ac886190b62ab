"""render_engine_amd: MI355X-native visible-set pipeline of render_engine (cull + ECS tick + instance pack).

The compute lives in lib/librender_engine_hip.so (hand-written HIP for gfx950 behind the C ABI of
include/re_hip.h).  There is no CPU fallback: using the package without the built library raises.
"""
from . import _capi  # noqa: F401
from .pipeline import Camera, Pipeline, RenderEngineError, ENTITY_DT, CHANGE_DT, create_level_of_views, section_keys, first_section_keys, shard_by_first_section  # noqa: F401
from ._capi import (F_STATIC, F_HAS_VEL, F_HAS_ACC, F_HAS_ROT, F_HAS_ROTVEL, F_HAS_ROTACC, F_HAS_SCALE,  # noqa: F401
                    F_ALWAYS_EXEC, F_OOB_LOGIC, F_HAS_MOVED, F_HAS_ROTATED, F_USER, F_CAN_COLLIDE, F_LIGHT_DIRECTIONAL, F_LIGHT_POINT, F_LIGHT_SPOT, F_PHANTOM)
from . import synthetic, lighting, history  # noqa: F401,E402   (parallel imports torch: import render_engine_amd.parallel where needed)

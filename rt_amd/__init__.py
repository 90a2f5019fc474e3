"""rt_amd — MI355X-native wavefront path tracer behind the render hot path of thabnir/rt.

`rt_amd.api` mirrors the reference's host API (Camera / World / Sphere / Triangle / materials /
textures); `rt_amd.scenes` holds the seeded synthetic scenes; all rendering goes through the C ABI of
`librt_hip.so` (include/rt_hip.h), built from rt_amd/csrc by `__graft_entry__.build()`.
Importing this package does not load the library; the first call that needs it does, and fails
loudly if it is missing.
"""
from . import _abi  # noqa: F401
from .api import (Camera, CheckerTexture, Dielectric, ImageTexture, Lambertian, Metal, SceneDesc,  # noqa: F401
                  SolidColor, Sphere, Triangle, World, encode_gamma8, generate_ground_plane, write_png, write_ppm)

__all__ = ["Camera", "World", "Sphere", "Triangle", "Lambertian", "Metal", "Dielectric", "SolidColor",
           "CheckerTexture", "ImageTexture", "SceneDesc", "generate_ground_plane", "encode_gamma8", "write_ppm", "write_png"]

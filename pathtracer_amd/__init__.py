"""pathtracer_amd -- MI355X-native backend for WeakKnight/PathTracer's ray-scene
intersection + Monte-Carlo integration loop."""
from . import capi, distributed, scene, scenes   # noqa: F401
from .pathtracer import DeviceScene, PathTracer, PathTracerGroup  # noqa: F401

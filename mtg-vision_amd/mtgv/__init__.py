"""mtgv - the recognition hot path of mtg-vision on MI355X (gfx950).

    from mtgv import Encoder, Detector, Matcher, Pipeline, spec
    from mtgv.adapters import CoreMlEncoder, CardSegmenter, VectorStoreQdrant   # the reference's names

Importing the package does not touch the GPU; constructing any of the classes does (and raises without one).
"""

from . import spec  # noqa: F401

_LAZY = {
    "Encoder": ("encoder", "Encoder"),
    "Detector": ("detector", "Detector"),
    "Detections": ("detector", "Detections"),
    "Matcher": ("matcher", "Matcher"),
    "merge_topk": ("matcher", "merge_topk"),
    "Pipeline": ("pipeline", "Pipeline"),
    "warp_quads": ("crop", "warp_quads"),
}


def __getattr__(name):
    if name in _LAZY:
        import importlib

        mod, attr = _LAZY[name]
        return getattr(importlib.import_module(f"{__name__}.{mod}"), attr)
    raise AttributeError(name)

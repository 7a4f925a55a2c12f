"""Import shim for the *reference* simfire package (container-only tooling).

The reference lives read-only at /root/reference and is never copied into this
repo.  It cannot be imported as-is here because pygame / noise / GIS wheels are
missing, so this module registers light stand-ins for those third-party
modules (display + I/O only, none of them takes part in the fire-spread
arithmetic) and exposes the real reference classes.  It is used ONLY by
``make_golden.py`` to generate the fixtures in this directory and by the
``reference`` timing script; nothing in the product or in the GPU tests
imports it (the reference does not exist on the GPU box).
"""
import os
import sys
import types
from unittest.mock import MagicMock

REFERENCE_ROOT = os.environ.get("SIMFIRE_REFERENCE", "/root/reference")


class _Rect:
    """Minimal pygame.Rect: only what Fire/Terrain sprites touch headless."""

    def __init__(self, x, y, w, h):
        self.x, self.y, self.w, self.h = int(x), int(y), int(w), int(h)

    def __iter__(self):
        return iter((self.x, self.y, self.w, self.h))

    def __getitem__(self, i):
        return (self.x, self.y, self.w, self.h)[i]

    def move(self, dx, dy):
        return _Rect(self.x + dx, self.y + dy, self.w, self.h)

    def update(self, x, y, w, h):
        self.x, self.y, self.w, self.h = int(x), int(y), int(w), int(h)


class _Sprite:
    def __init__(self, *a, **k):
        pass


def install():
    if "simfire" in sys.modules and getattr(sys.modules["simfire"], "_is_ref", False):
        return
    if not os.path.isdir(os.path.join(REFERENCE_ROOT, "simfire")):
        raise RuntimeError(f"reference not found at {REFERENCE_ROOT}")
    pg = MagicMock(name="pygame")
    pg.Rect = _Rect
    pg.rect = types.SimpleNamespace(Rect=_Rect)
    pg.sprite = types.SimpleNamespace(Sprite=_Sprite)
    sys.modules["pygame"] = pg
    sys.modules["pygame.sprite"] = pg.sprite
    sys.modules["pygame.rect"] = pg.rect
    sys.modules["pygame.surface"] = MagicMock()
    sys.modules["pygame.surfarray"] = MagicMock()
    for name in [
        "noise", "skimage", "skimage.draw", "wurlitzer", "reportlab",
        "reportlab.graphics", "svglib", "svglib.svglib", "h5py", "jsonlines",
        "cv2", "imagecodecs", "landfire", "landfire.product", "landfire.product.enums",
        "landfire.product.search", "geotiff", "geopandas", "geopy", "geopy.distance",
    ]:
        sys.modules.setdefault(name, MagicMock(name=name))
    pkg = types.ModuleType("simfire")
    pkg.__path__ = [os.path.join(REFERENCE_ROOT, "simfire")]
    pkg._is_ref = True
    sys.modules["simfire"] = pkg
    os.environ.setdefault("SDL_VIDEODRIVER", "dummy")


install()

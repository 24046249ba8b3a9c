"""Drop-in module name for scripts written against the reference renderers: ``from T3DR import Photon``.
``Photon`` / ``Rendered`` live in ``raytracing_amd.photon``.  The rasteriser preview (``Texel``), the
orthographic debug viewer (``Debug``) of the reference are not part of this build; the names exist so that
scripts which merely construct them keep importing.  ``Post`` (image helpers) is in ``raytracing_amd.post``."""
from raytracing_amd.photon import NoBoundCameraError, Photon, Rendered, ThreadsProgress  # noqa: F401
from raytracing_amd.post import Post  # noqa: F401


class _NotBuilt:
    _what = "engine"

    def __init__(self, scene=None, options=None):
        self.scene = scene
        self.options = {} if options is None else options

    def __getattr__(self, name):
        raise NotImplementedError(f"T3DR.{type(self).__name__}.{name}: the {self._what} is outside the scope of "
                                  "the MI355X build (only the Photon ray tracer is implemented)")


class Debug(_NotBuilt):
    _what = "orthographic debug viewer"


class Texel(_NotBuilt):
    _what = "rasteriser preview"


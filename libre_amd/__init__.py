"""libre_amd -- MI355X-native volume raycaster behind Libre's renderer-plugin interface.

Layout:
  csrc/   hand-written HIP kernels for gfx950 + the C ABI (include/vrc_hip.h) -> lib/libvrc_hip.so
  host/   C++17 mirror of Libre's plugin surface (RendererPlugin, RenderPipelinePlugin,
          Cache, TextureObject, TexturePool, DataSource) -> lib/libLivreHipRaycastPipeline.so
  vrc.py  ctypes binding of the C ABI (what a cgo/JNI/ctypes integrator would write)

There is no CPU fallback: loading fails loudly if the HIP library has not been built.
"""
from .vrc import VrcError, load_library  # noqa: F401

"""gnumap_amd — Python binding (ctypes) of libgnumap_hip.so, the MI355X (gfx950) implementation of GNUMAP's
per-read seed-and-extend hot path.  The library is the product; this module only loads it and mirrors the C ABI
declared in include/gnumap_hip.h.  There is no CPU fallback: compute calls raise GnumapError when no gfx950
device (or no built library) is available."""
from .api import (GnumapError, Index, Batch, Params, lib, load_library, library_path, pack_reads, GM_INDEX_FULL_SA, GM_INDEX_BUILD,
                  GM_INDEX_HOST_ONLY, GM_READ_OK, GM_READ_TOO_MANY, GM_READ_NONE, GM_READ_TOO_SHORT, GM_READ_TOO_POOR,
                  GM_BUILD_AUTO, GM_BUILD_HOST, GM_BUILD_DEVICE, index_build, version, set_option)

__all__ = ["GnumapError", "Index", "Batch", "Params", "lib", "load_library", "library_path", "pack_reads", "index_build", "version", "set_option",
           "GM_INDEX_FULL_SA", "GM_INDEX_BUILD", "GM_INDEX_HOST_ONLY", "GM_BUILD_AUTO", "GM_BUILD_HOST", "GM_BUILD_DEVICE",
           "GM_READ_OK", "GM_READ_TOO_MANY", "GM_READ_NONE", "GM_READ_TOO_SHORT", "GM_READ_TOO_POOR"]

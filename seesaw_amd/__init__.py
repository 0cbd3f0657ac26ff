"""seesaw_amd -- MI355X (gfx950) implementation of seesaw's interactive-search hot path.

Host code is Python mirroring the reference's own interface (seesaw.query_interface,
seesaw.vector_index, seesaw.seesaw_bench, the indices and the feedback loops); every
numeric step runs in libseesaw_hip.so (hand-written HIP, C-ABI in include/seesaw_hip.h).
There is no CPU fallback: importing a module that needs the library raises if it has
not been built.
"""
__version__ = "0.1.0"

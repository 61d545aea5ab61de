"""Stub of DI-engine: only what /root/reference/lzero/model/*.py imports (see ../README.md)."""

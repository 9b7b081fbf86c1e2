"""vit.cpp_amd -- MI355X-native drop-in for staghado/vit.cpp's forward path.

The product is the C-ABI shared library built from csrc/ (include/vitx.h) plus the
C++ vit.h mirror; this Python package is plumbing for tests and bench.py: the
file-format writer, synthetic weights, and a ctypes binding over the C ABI.
Import it through the repo-root helper: `import _pkg; vit = _pkg.load()`.
"""
from . import convert, dist, ggml_file, synth  # noqa: F401

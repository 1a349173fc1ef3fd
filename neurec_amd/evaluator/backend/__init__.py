"""Backend selection.  The reference tries its C++ evaluator and silently falls back to a
pure-Python one (evaluator/backend/__init__.py:1-6).  Here there is exactly one backend — the
HIP kernels — and a missing/unloadable library is an import error, never a fallback."""
from .hip.uni_evaluator import UniEvaluator

print("Evaluate model with hip (gfx950)")

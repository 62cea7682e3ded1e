"""hisat2_amd — MI355X-native seed-and-extend hot path behind HISAT2's call boundary.

Only the path of SURVEY.md §8 lives here: the C-ABI library (csrc/ -> libh2g.so, hand-written
HIP for gfx950) and the thin host-side mirror used by tests and bench.py.
"""
__all__ = ["api", "synth"]

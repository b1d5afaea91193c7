"""MI355X-native ThinkDiff hot path, exposed under the reference's import path (`thinkdiff.*`).

Reference: avi22bhattacharya/ThinkDiff-mlre `thinkdiff/__init__.py` registers paths and imports the
model/runner/task registries; here the package only wires the HIP library and the thin mirrors of
the reference entry points that sit on the inference hot path.
"""
__all__ = ["_hip"]

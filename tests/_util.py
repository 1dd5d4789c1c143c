"""Small helpers shared by the GPU tests: self-describing comparisons (a failure names the field, the count and the place
of the differences -- never a dump of the arrays)."""
import hashlib
import json
import os

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PARITY_REPORT = os.path.join(_ROOT, "gpurun_out", "parity_report.json")
_parity = {}


def parity_record(test, **values):
    """Record what a parity test ACHIEVED (max errors, fills, counts) under its name and rewrite
    gpurun_out/parity_report.json -- the numbers behind the asserts, kept with the run (a summary is committed under
    profiles/).  Values: floats / ints / strings."""
    entry = _parity.setdefault(test, {})
    for k, v in values.items():
        entry[k] = v if isinstance(v, (str, bool)) else (int(v) if isinstance(v, (int, np.integer)) else float(v))
    try:
        os.makedirs(os.path.dirname(PARITY_REPORT), exist_ok=True)
        old = {}
        if os.path.exists(PARITY_REPORT):
            with open(PARITY_REPORT) as fh:
                old = json.load(fh)
        old.update(_parity)
        with open(PARITY_REPORT, "w") as fh:
            json.dump(old, fh, indent=1, sort_keys=True)
    except (OSError, ValueError):
        pass  # the report is a by-product: never the reason a parity test fails


def digest(a):
    """short sha1 of an array's bytes"""
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def bits_report(a, b, name="array"):
    """'' when a and b are bit-identical, else one line: digests, number of differing entries, the first differing index
    with both values (hex), the largest relative difference."""
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    if a.shape != b.shape:
        return f"{name}: shapes differ {a.shape} vs {b.shape}"
    ua, ub = a.view(np.uint8).reshape(a.size, -1), b.view(np.uint8).reshape(b.size, -1)
    diff = np.flatnonzero(np.any(ua != ub, axis=1))
    if diff.size == 0:
        return ""
    i = int(diff[0])
    fa, fb = a.ravel()[i], b.ravel()[i]
    scale = np.maximum(np.abs(a.ravel()), np.abs(b.ravel()))
    with np.errstate(invalid="ignore", divide="ignore"):
        rel = np.abs(a.ravel() - b.ravel()) / np.where(scale > 0, scale, 1.0)
    hx = (lambda v: float(v).hex()) if a.dtype.kind == "f" else repr
    return (f"{name}: NOT bit-identical -- digests {digest(a)} vs {digest(b)}; {diff.size} of {a.size} entries differ; first at "
            f"[{i}]: {hx(fa)} vs {hx(fb)}; max relative difference {np.nanmax(rel):.3e}"
            f"{'; non-finite values present' if not (np.all(np.isfinite(a)) and np.all(np.isfinite(b))) else ''}")


def rel_err(a, b):
    nb = np.linalg.norm(b)
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / (nb if nb > 0 else 1.0))

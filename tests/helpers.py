"""Helpers shared by the test modules."""
import os

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def rel_err(a, b):
    """max|a-b| / max(|b|, tiny): norm-relative error used for H, b and poses."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.size == 0:
        return 0.0
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def b_err(b, b_ref, H_ref, n_inliers, error):
    """Error of b = sum J^T e relative to the magnitude of its terms.

    b cancels to ~0 at the optimum, so |b| itself is not a usable scale; by Cauchy-Schwarz
    |b_i| <= sqrt(H_ii * sum(chi)) and sum(chi) = n_inliers * error, which is the size of the
    terms being summed.
    """
    b = np.asarray(b, dtype=np.float64)
    b_ref = np.asarray(b_ref, dtype=np.float64)
    if not np.isfinite(error) or n_inliers == 0:
        return float(np.max(np.abs(b - b_ref)))
    scale = np.sqrt(np.max(np.diag(H_ref)) * n_inliers * error)
    scale = max(scale, np.max(np.abs(b_ref)), 1e-300)
    return float(np.max(np.abs(b - b_ref)) / scale)


def entry_err(H, H_ref):
    """Largest error of an entry of H = sum J^T J relative to that entry's own scale sqrt(H_ii H_jj)
    (|H_ij| <= sqrt(H_ii H_jj) by Cauchy-Schwarz).  H's translation and rotation blocks differ by orders
    of magnitude, so a norm-relative check would let a small block be wrong at far more than the bar."""
    H = np.asarray(H, dtype=np.float64).reshape(6, 6)
    H_ref = np.asarray(H_ref, dtype=np.float64).reshape(6, 6)
    d = np.sqrt(np.abs(np.diag(H_ref)))
    scale = np.outer(d, d)
    if not np.all(scale > 0):
        return rel_err(H, H_ref) if np.max(np.abs(H_ref)) > 0 else float(np.max(np.abs(H)))
    return float(np.max(np.abs(H - H_ref) / scale))

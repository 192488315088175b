"""Kneedle elbow detection for the ``max_iter=None`` warm-up of ``ALPINE.fit`` (alpine/main.py:116-129, :755-770).

The reference delegates to the third-party package ``kneed`` (pinned ``kneed>=0.8.5,<0.9`` in its pyproject.toml:18),
calling ``KneeLocator(x, log10(recon loss), curve="convex", direction="decreasing", interp_method="polynomial",
polynomial_degree=2).elbow`` (main.py:758-765).  ``kneed`` is not vendored in the reference and is not installed in
this image, so ``ALPINE._compute_best_iter`` uses the real package whenever it is importable and otherwise this
restatement of the published algorithm (Satopaa, Albrecht, Irwin, Raghavan: "Finding a 'Kneedle' in a Haystack",
ICDCS-W 2011) in the form kneed 0.8 implements it (offline mode, sensitivity S = 1):

  1. smooth: least-squares polynomial of the given degree through (x, y);
  2. normalise x and the smoothed y to [0, 1];
  3. turn the (convex, decreasing) elbow into a knee: y <- max(y) - y;
  4. difference curve d = y - x; its local maxima (>=) are knee candidates, local minima (<=) reset the threshold;
  5. threshold of a candidate = d(candidate) - S * mean(|dx|);
  6. walk the curve from the first candidate; the first point where d drops below the current threshold confirms
     the candidate -> elbow = x[candidate].

PARITY UNPINNED against kneed for the reference's configuration: neither the reference's tests (it has none) nor anything
runnable here (kneed is absent, nothing may be installed) pins the elbow index this returns against kneed's.  What IS
pinned is the core of the algorithm (steps 2, 4-6) by the worked example of the Kneedle manuscript itself (its Figure 2:
y = -1/(x + 0.1) + 5 on ten points of [0, 1], concave and increasing, S = 1 -> knee at x = 0.22):
``find_knee(x, y, curve="concave", direction="increasing")`` returns 0.2222 (tests/test_host_api.py), and the four
(curve, direction) transforms by the ten-point sample curves of kneed's own DataGenerator with the knees its test-suite
states for them (2, 7, 7, 2; written down from the published tests, kneed itself cannot run here); the rest of the tests
cover invariants.
"""
from __future__ import annotations

from typing import Optional

import numpy as np


def _local_extrema(d: np.ndarray, greater: bool) -> np.ndarray:
    """Indices i (interior) with d[i] >= both neighbours (or <=): scipy.signal.argrelextrema(d, np.greater_equal)
    with order 1 and mode 'clip' also reports boundary points that tie with their single neighbour's clipped copy."""
    n = len(d)
    left = np.concatenate(([d[0]], d[:-1]))
    right = np.concatenate((d[1:], [d[-1]]))
    mask = (d >= left) & (d >= right) if greater else (d <= left) & (d <= right)
    return np.flatnonzero(mask)


def _kneedle_walk(x: np.ndarray, xn: np.ndarray, d: np.ndarray, S: float) -> Optional[float]:
    """Steps 4-6 on a difference curve d over normalised abscissae xn (knee form: concave, increasing)."""
    maxima = _local_extrema(d, greater=True)
    minima = _local_extrema(d, greater=False)
    if maxima.size == 0:
        return None
    tmx = d[maxima] - S * np.abs(np.diff(xn).mean())
    threshold, threshold_index, mi = 0.0, int(maxima[0]), 0
    is_max = np.zeros(len(d), dtype=bool); is_max[maxima] = True
    is_min = np.zeros(len(d), dtype=bool); is_min[minima] = True
    for i in range(int(maxima[0]), len(d)):
        if xn[i] == 1.0:
            break
        if is_max[i]:
            threshold = tmx[mi]
            threshold_index = i
            mi += 1
        if is_min[i]:
            threshold = 0.0
        if d[i + 1] < threshold:
            return float(x[threshold_index])
    return None


def find_knee(x, y, curve: str = "concave", direction: str = "increasing", S: float = 1.0,
              polynomial_degree: Optional[int] = None) -> Optional[float]:
    """Knee / elbow of a curve in the four (curve, direction) forms of the Kneedle manuscript; ``polynomial_degree`` = None
    uses the points as they are (the manuscript's worked example), an integer smooths with a least-squares polynomial."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    if len(x) < 3 or not np.all(np.isfinite(y)):
        return None
    ds_y = y if polynomial_degree is None else np.poly1d(np.polyfit(x, y, polynomial_degree))(x)
    span_x, span_y = x.max() - x.min(), ds_y.max() - ds_y.min()
    if span_x == 0 or span_y == 0:
        return None
    xn = (x - x.min()) / span_x
    yn = (ds_y - ds_y.min()) / span_y
    # bring every form to the knee form (concave, increasing), as kneed's transform_y does
    if curve == "convex" and direction == "decreasing":
        yn = yn.max() - yn
    elif curve == "convex" and direction == "increasing":
        xn_, yn_ = xn.max() - xn[::-1], yn.max() - yn[::-1]
        knee = _kneedle_walk(x[::-1], xn_, yn_ - xn_, S)
        return knee
    elif curve == "concave" and direction == "decreasing":
        xn_, yn_ = xn.max() - xn[::-1], yn[::-1]
        return _kneedle_walk(x[::-1], xn_, yn_ - xn_, S)
    elif not (curve == "concave" and direction == "increasing"):
        raise ValueError("curve must be 'concave' or 'convex', direction 'increasing' or 'decreasing'")
    return _kneedle_walk(x, xn, yn - xn, S)


def find_elbow(x, y, S: float = 1.0, polynomial_degree: int = 2) -> Optional[float]:
    """Elbow of a convex, decreasing curve (the reference's call, main.py:758-765); returns the x value or None."""
    return find_knee(x, y, curve="convex", direction="decreasing", S=S, polynomial_degree=polynomial_degree)

"""Feature-set comparison used by the parity tests and the bench's self-check.

The reference orders its output with an unstable sort on scale, so feature lists are compared as
sets: every reference feature is paired with the nearest unused candidate that lies within the
tolerances (position in px, relative scale, orientation in degrees).
"""
import numpy as np


def _ang_diff_deg(a, b):
    d = np.abs(a - b) % (2 * np.pi)
    return np.degrees(np.minimum(d, 2 * np.pi - d))


def match_features(ref, got, pos_tol=0.5, scale_tol=0.01, ori_tol_deg=1.0):
    """ref, got: structured arrays with dx, dy, scale, ori. Returns index pairs (ri, gi)."""
    if len(ref) == 0 or len(got) == 0:
        return np.zeros((0, 2), dtype=np.int64)
    # bucket candidates on a 1-px grid
    cell = {}
    gx, gy = np.asarray(got["dx"], dtype=np.float64), np.asarray(got["dy"], dtype=np.float64)
    for gi in range(len(got)):
        cell.setdefault((int(np.floor(gx[gi])), int(np.floor(gy[gi]))), []).append(gi)
    used = np.zeros(len(got), dtype=bool)
    pairs = []
    for ri in range(len(ref)):
        rx, ry, rs, ro = float(ref["dx"][ri]), float(ref["dy"][ri]), float(ref["scale"][ri]), float(ref["ori"][ri])
        cx, cy = int(np.floor(rx)), int(np.floor(ry))
        best, best_d = -1, None
        for ox in (-1, 0, 1):
            for oy in (-1, 0, 1):
                for gi in cell.get((cx + ox, cy + oy), ()):
                    if used[gi]:
                        continue
                    d = np.hypot(gx[gi] - rx, gy[gi] - ry)
                    if d > pos_tol:
                        continue
                    if abs(float(got["scale"][gi]) - rs) > scale_tol * rs:
                        continue
                    if _ang_diff_deg(float(got["ori"][gi]), ro) > ori_tol_deg:
                        continue
                    key = (d, abs(float(got["scale"][gi]) - rs), _ang_diff_deg(float(got["ori"][gi]), ro))
                    if best_d is None or key < best_d:
                        best, best_d = gi, key
        if best >= 0:
            used[best] = True
            pairs.append((ri, best))
    return np.asarray(pairs, dtype=np.int64).reshape(-1, 2)


def descriptor_agreement(ref_desc, got_desc):
    """ref_desc, got_desc: (n,128) integer-valued arrays of matched features.
    Returns (fraction of identical descriptors, fraction of identical bins, max normalised L2)."""
    if len(ref_desc) == 0:
        return 1.0, 1.0, 0.0
    r = np.asarray(ref_desc, dtype=np.float64)
    g = np.asarray(got_desc, dtype=np.float64)
    same = r == g
    rn = r / np.maximum(np.linalg.norm(r, axis=1, keepdims=True), 1e-300)
    gn = g / np.maximum(np.linalg.norm(g, axis=1, keepdims=True), 1e-300)
    l2 = np.linalg.norm(rn - gn, axis=1)
    return float(same.all(axis=1).mean()), float(same.mean()), float(l2.max())

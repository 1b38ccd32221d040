"""numpy restatement of the 12-relation spatial graph.

TEST INFRASTRUCTURE — see oracle/__init__.py.  Follows /root/reference/sam/spatial_utils.py:92-218
(pair classification), :55-89 (shared-sector replace maps), :33-52 (one-hot broadcast) and
/root/reference/sam/datasets/textvqa_dataset.py:378-409 (composition into c=3/5/7/9 matrices).
Pinned by the reference's known-answer vector and by goldens generated from the
reference itself (tests/golden/spatial_graph.npz).

Relation codes: 0 none, 1 i covers j, 2 i inside j, 3 IoU>=0.5, 4..11 eight 45-degree sectors of
the centre-to-centre direction (only if the centre distance < threshold*sqrt(2)), 12 self.
"""
import math

import numpy as np

SHARE_KEYS = ("1", "31", "32", "51", "52", "71", "72", "91", "92")
BUILD_MAP = {"3": ("1", "31", "32"), "5": ("3", "51", "52"), "7": ("5", "71", "72"), "9": ("7", "91", "92")}


def replace_maps():
    """sector -> neighbouring sector at distance +-k with wrap-around in 4..11 (spatial_utils.py:55-89)."""
    maps = {"1": {}}
    for width, k in (("3", 1), ("5", 2), ("7", 3), ("9", 4)):
        maps[width + "1"] = {s: 4 + (s - 4 + k) % 8 for s in range(4, 12)}
        maps[width + "2"] = {s: 4 + (s - 4 - k) % 8 for s in range(4, 12)}
    return maps


def _iou(a, b):
    """spatial_utils.py:7-30."""
    iw = max(0, min(a[2], b[2]) - max(a[0], b[0]))
    ih = max(0, min(a[3], b[3]) - max(a[1], b[1]))
    inter = iw * ih
    return inter / float((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter)


def _sector_pair(dy, dx):
    """codes (i->j, j->i) for centre offset (dy, dx) = centre_i - centre_j (spatial_utils.py:168-203)."""
    dist = math.sqrt(dy * dy + dx * dx)
    if dist == 0.0:  # reference: 0/0 -> nan -> both ceil() are nan -> code 4 for both directions
        return 4, 4
    s, c = dy / dist, dx / dist
    if s >= 0 and c >= 0:
        li = float(np.arcsin(s)); lj = math.pi + li
    elif s < 0 and c >= 0:
        li = float(np.arcsin(s)) + 2 * math.pi; lj = li - math.pi
    elif s >= 0 and c < 0:
        li = float(np.arccos(c)); lj = li + math.pi
    else:
        li = 2 * math.pi - float(np.arccos(c)); lj = li - math.pi
    q = math.pi / 4
    return int(np.ceil(li / q)) + 3, int(np.ceil(lj / q)) + 3


def relation_codes(bbox, distance_threshold=0.5):
    """dict key -> int8 [N,N] code matrix; O(N^2) scalar loop, the readable form."""
    bbox = np.asarray(bbox, dtype=np.float64)
    n = bbox.shape[0]
    maps = replace_maps()
    out = {k: np.zeros((n, n), dtype=np.int64) for k in SHARE_KEYS}
    cx, cy = 0.5 * (bbox[:, 0] + bbox[:, 2]), 0.5 * (bbox[:, 1] + bbox[:, 3])
    limit = distance_threshold * math.sqrt(2.0)
    for i in range(n):
        if bbox[i].sum() == 0:
            continue
        out["1"][i, i] = 12
        for j in range(i + 1, n):
            if bbox[j].sum() == 0:
                continue
            a, b = bbox[i], bbox[j]
            if a[0] < b[0] and a[2] > b[2] and a[1] < b[1] and a[3] > b[3]:
                out["1"][i, j], out["1"][j, i] = 1, 2
            elif b[0] < a[0] and b[2] > a[2] and b[1] < a[1] and b[3] > a[3]:
                out["1"][i, j], out["1"][j, i] = 2, 1
            elif _iou(a, b) >= 0.5:
                out["1"][i, j] = out["1"][j, i] = 3
            else:
                dy, dx = cy[i] - cy[j], cx[i] - cx[j]
                if math.sqrt(dy * dy + dx * dx) < limit:
                    cij, cji = _sector_pair(dy, dx)
                    out["1"][i, j], out["1"][j, i] = cij, cji
                    for k in SHARE_KEYS[1:]:
                        out[k][i, j] = maps[k].get(cij, 0)
                        out[k][j, i] = maps[k].get(cji, 0)
    return {k: v.astype(np.int8) for k, v in out.items()}


def multi_hot(code):
    """code r in 1..12 -> one-hot channel r-1, 0 -> all zero (spatial_utils.py:33-52). int8 [N,N,12]."""
    code = np.asarray(code)
    out = np.zeros(code.shape + (12,), dtype=np.int8)
    ii, jj = np.nonzero(code > 0)
    out[ii, jj, code[ii, jj].astype(np.int64) - 1] = 1
    return out


def compose(codes, context):
    """relation tensor for spatial context c in {1,3,5,7,9} (textvqa_dataset.py:378-409). int8 [N,N,12]."""
    mats = {"1": multi_hot(codes["1"])}
    for c in ("3", "5", "7", "9"):
        base, plus, minus = BUILD_MAP[c]
        mats[c] = np.maximum(np.maximum(mats[base], multi_hot(codes[plus])), multi_hot(codes[minus]))
        if c == str(context):
            break
    return mats[str(context)]

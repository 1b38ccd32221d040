"""Spatial relation graph builder, vectorised over all box pairs and batched, running on
whatever device the boxes live on (the GPU in training; the reference runs an O(N^2) Python loop at 0.49 s/sample).

Semantics follow /root/reference/sam/spatial_utils.py:92-218 (pair classification; per pair (i<j) the i->j sector is
computed from centre_i - centre_j and the j->i sector as that angle +- pi), :55-89 (shared-sector maps), :33-52
(one-hot broadcast) and sam/datasets/textvqa_dataset.py:378-409 (context composition).  All arithmetic is float64 in the
reference's operation order, so codes are bit-identical up to libm asin/acos last-ulp differences at exact sector
boundaries (parity-tested against the oracle and the reference goldens).
"""
import math

import torch

_CTX_WIDTH = {1: 0, 3: 1, 5: 2, 7: 3, 9: 4}


def relation_codes(boxes, distance_threshold=0.5):
    """boxes: float [B, N, 4] normalised xyxy, all-zero rows = padding -> int8 [B, N, N] base relation codes ("1" matrix)"""
    bx = boxes.to(torch.float64)
    B, N, _ = bx.shape
    x0, y0, x1, y1 = bx.unbind(-1)
    valid = bx.sum(-1) != 0
    pair_valid = valid[:, :, None] & valid[:, None, :]

    def a_(t):  # row index a
        return t[:, :, None]

    def b_(t):  # column index b
        return t[:, None, :]

    covers = (a_(x0) < b_(x0)) & (a_(x1) > b_(x1)) & (a_(y0) < b_(y0)) & (a_(y1) > b_(y1))       # a covers b
    iw = (torch.minimum(a_(x1), b_(x1)) - torch.maximum(a_(x0), b_(x0))).clamp(min=0)
    ih = (torch.minimum(a_(y1), b_(y1)) - torch.maximum(a_(y0), b_(y0))).clamp(min=0)
    inter = iw * ih
    area = (x1 - x0) * (y1 - y0)
    iou = inter / ((a_(area) + b_(area)) - inter)
    # sector of the pair (i=min(a,b), j=max(a,b)): diff = centre_i - centre_j
    cx, cy = 0.5 * (x0 + x1), 0.5 * (y0 + y1)
    upper = torch.ones(N, N, dtype=torch.bool, device=bx.device).triu(1)[None]                 # a < b
    dy = torch.where(upper, a_(cy) - b_(cy), b_(cy) - a_(cy))
    dx = torch.where(upper, a_(cx) - b_(cx), b_(cx) - a_(cx))
    dist = torch.sqrt(dy * dy + dx * dx)
    s, c = dy / dist, dx / dist
    asin_s, acos_c = torch.asin(s.clamp(-1, 1)), torch.acos(c.clamp(-1, 1))
    pi = math.pi
    q1, q4, q2 = (s >= 0) & (c >= 0), (s < 0) & (c >= 0), (s >= 0) & (c < 0)
    li = torch.where(q1, asin_s, torch.where(q4, asin_s + 2 * pi, torch.where(q2, acos_c, 2 * pi - acos_c)))
    lj = torch.where(q1, pi + li, torch.where(q4, li - pi, torch.where(q2, li + pi, li - pi)))
    lab = torch.where(upper, li, lj)                                                           # direction a -> b
    sector = torch.ceil(lab / (pi / 4)) + 3
    sector = torch.where(torch.isnan(sector), torch.full_like(sector, 4.0), sector)            # coincident centres: nan -> 4
    near = dist < distance_threshold * math.sqrt(2.0)
    near = near | torch.isnan(dist)
    code = torch.where(near, sector, torch.zeros_like(sector))
    code = torch.where(iou >= 0.5, torch.full_like(code, 3.0), code)
    code = torch.where(covers.transpose(1, 2), torch.full_like(code, 2.0), code)
    code = torch.where(covers, torch.full_like(code, 1.0), code)
    code = torch.where(pair_valid, code, torch.zeros_like(code))
    eye = torch.eye(N, dtype=torch.bool, device=bx.device)[None]
    code = torch.where(eye, torch.where(valid, 12.0, 0.0)[:, :, None].expand(B, N, N).to(code.dtype), code)
    return code.to(torch.int8)


def relation_tensor(boxes, context=3, distance_threshold=0.5):
    """int8 multi-hot [B, N, N, 12] for spatial context c in {1,3,5,7,9}: channel r-1 set where the pair's relation is r,
    plus, for sector relations, the neighbouring sectors within +-(c-1)/2 (wrapping inside 4..11)."""
    code = relation_codes(boxes, distance_threshold).to(torch.int64)
    B, N, _ = code.shape
    out = torch.zeros(B, N, N, 12, dtype=torch.int8, device=code.device)
    nz = code > 0
    out.scatter_(3, (code - 1).clamp(min=0).unsqueeze(-1), nz.to(torch.int8).unsqueeze(-1))
    is_sector = (code >= 4) & (code <= 11)
    for k in range(1, _CTX_WIDTH[int(context)] + 1):
        for sgn in (1, -1):
            nb = 4 + (code - 4 + sgn * k) % 8
            idx = torch.where(is_sector, nb - 1, torch.zeros_like(nb)).unsqueeze(-1)
            cur = out.gather(3, idx)
            out.scatter_(3, idx, torch.maximum(cur, is_sector.to(torch.int8).unsqueeze(-1)))
    return out

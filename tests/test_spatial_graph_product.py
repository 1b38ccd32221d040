"""CPU: the product-side vectorised graph builder equals the oracle (and hence the reference goldens) bit for bit."""
import numpy as np
import torch

from oracle import spatial_graph as SG
from tests import oracle_cases as OC
from tests.golden import common as C


def test_vectorised_builder_matches_reference_goldens():
    from sam_textvqa_amd.spatial_graph import relation_codes, relation_tensor
    g = OC.load("spatial_graph")
    for nm in ("known6", "grid", "rnd60", "cross"):
        boxes = torch.from_numpy(g[nm + ".boxes"])[None]
        np.testing.assert_array_equal(relation_codes(boxes)[0].numpy(), g[nm + ".code1"], err_msg=nm)
        for ctx in (1, 3, 5, 7, 9):
            np.testing.assert_array_equal(relation_tensor(boxes, ctx)[0].numpy(), g["%s.ctx%d" % (nm, ctx)], err_msg="%s ctx%d" % (nm, ctx))


def test_vectorised_builder_matches_oracle_batched():
    from sam_textvqa_amd.spatial_graph import relation_tensor
    boxes = np.stack([np.concatenate([C.det_boxes("sgp%d.o" % b, 100 - 7 * b, 100, 0.21), C.det_boxes("sgp%d.t" % b, 50 - 20 * b, 50, 0.08)])
                      for b in range(3)])
    for ctx in (3, 5):
        got = relation_tensor(torch.from_numpy(boxes), ctx).numpy()
        for b in range(3):
            with np.errstate(all="ignore"):
                np.testing.assert_array_equal(got[b], SG.compose(SG.relation_codes(boxes[b]), ctx))

"""Layer table of the XFeat network (host-side mirror of the reference interface).

Names are the reference's ``state_dict`` prefixes so a reference checkpoint loads
unchanged (reference: modules/model.py:33-111, key list SURVEY.md App. A.1).

Two kinds of convolution exist on the path:

* ``bn``    -- "BasicLayer": Conv2d(bias=False) -> BatchNorm2d(affine=False, eval) -> ReLU
               (reference: modules/model.py:12-25).  Keys ``<name>.layer.0.weight``,
               ``<name>.layer.1.running_mean|running_var|num_batches_tracked``.
* ``plain`` -- Conv2d with bias, no norm, no activation (modules/model.py:41,76,82,91).
               Keys ``<name>.weight``, ``<name>.bias``.
"""
from collections import namedtuple

BN_EPS = 1e-5          # nn.BatchNorm2d / BatchNorm1d default eps
IN_EPS = 1e-5          # nn.InstanceNorm2d default eps (modules/model.py:35)

Conv = namedtuple("Conv", "name cin cout k stride kind")

# Order == execution order of XFeatModel.forward (modules/model.py:123-154).
CONVS = [
    Conv("skip1.1",          1,  24, 1, 1, "plain"),   # after AvgPool2d(4,4)
    Conv("block1.0",         1,   4, 3, 1, "bn"),
    Conv("block1.1",         4,   8, 3, 2, "bn"),
    Conv("block1.2",         8,   8, 3, 1, "bn"),
    Conv("block1.3",         8,  24, 3, 2, "bn"),
    Conv("block2.0",        24,  24, 3, 1, "bn"),
    Conv("block2.1",        24,  24, 3, 1, "bn"),
    Conv("block3.0",        24,  64, 3, 2, "bn"),
    Conv("block3.1",        64,  64, 3, 1, "bn"),
    Conv("block3.2",        64,  64, 1, 1, "bn"),
    Conv("block4.0",        64,  64, 3, 2, "bn"),
    Conv("block4.1",        64,  64, 3, 1, "bn"),
    Conv("block4.2",        64,  64, 3, 1, "bn"),
    Conv("block5.0",        64, 128, 3, 2, "bn"),
    Conv("block5.1",       128, 128, 3, 1, "bn"),
    Conv("block5.2",       128, 128, 3, 1, "bn"),
    Conv("block5.3",       128,  64, 1, 1, "bn"),
    Conv("block_fusion.0",  64,  64, 3, 1, "bn"),
    Conv("block_fusion.1",  64,  64, 3, 1, "bn"),
    Conv("block_fusion.2",  64,  64, 1, 1, "plain"),
    Conv("heatmap_head.0",  64,  64, 1, 1, "bn"),
    Conv("heatmap_head.1",  64,  64, 1, 1, "bn"),
    Conv("heatmap_head.2",  64,   1, 1, 1, "plain"),   # followed by Sigmoid
    Conv("keypoint_head.0", 64,  64, 1, 1, "bn"),
    Conv("keypoint_head.1", 64,  64, 1, 1, "bn"),
    Conv("keypoint_head.2", 64,  64, 1, 1, "bn"),
    Conv("keypoint_head.3", 64,  65, 1, 1, "plain"),
]
CONV_BY_NAME = {c.name: c for c in CONVS}
CONV_INDEX = {c.name: i for i, c in enumerate(CONVS)}

# fine_matcher MLP (modules/model.py:97-111): Linear(+bias) -> BatchNorm1d(affine=False)
# -> ReLU, four times, then a final Linear.  (index of Linear, fan_in, fan_out, index of BN)
FINE = [
    (0, 128, 512, 1),
    (3, 512, 512, 4),
    (6, 512, 512, 7),
    (9, 512, 512, 10),
    (12, 512, 64, None),
]


def state_dict_keys():
    """All keys (and shapes) a reference XFeatModel state_dict holds."""
    out = {}
    for c in CONVS:
        if c.kind == "bn":
            out[f"{c.name}.layer.0.weight"] = (c.cout, c.cin, c.k, c.k)
            out[f"{c.name}.layer.1.running_mean"] = (c.cout,)
            out[f"{c.name}.layer.1.running_var"] = (c.cout,)
            out[f"{c.name}.layer.1.num_batches_tracked"] = ()
        else:
            out[f"{c.name}.weight"] = (c.cout, c.cin, c.k, c.k)
            out[f"{c.name}.bias"] = (c.cout,)
    for li, fin, fout, bi in FINE:
        out[f"fine_matcher.{li}.weight"] = (fout, fin)
        out[f"fine_matcher.{li}.bias"] = (fout,)
        if bi is not None:
            out[f"fine_matcher.{bi}.running_mean"] = (fout,)
            out[f"fine_matcher.{bi}.running_var"] = (fout,)
            out[f"fine_matcher.{bi}.num_batches_tracked"] = ()
    return out

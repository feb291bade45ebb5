"""accelerated_features_amd -- the XFeat inference hot path, hand-written for MI355X (gfx950).

    from accelerated_features_amd import XFeat      # drop-in for modules.xfeat.XFeat
"""
from .xfeat import XFeat, XFeatModel  # noqa: F401

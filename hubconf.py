"""torch.hub entry point, same signature as the reference's hubconf.py
(/root/reference/hubconf.py:5-15)."""
dependencies = ['torch']
import torch

from accelerated_features_amd.xfeat import XFeat as _XFeat


def XFeat(pretrained=True, top_k=4096, detection_threshold=0.05):
    """
    XFeat model (MI355X-native)
    pretrained (bool): kwargs, load pretrained weights into the model
    """
    weights = None
    if pretrained:
        weights = torch.hub.load_state_dict_from_url(
            "https://github.com/verlab/accelerated_features/raw/main/weights/xfeat.pt", map_location='cpu')
    return _XFeat(weights, top_k=top_k, detection_threshold=detection_threshold)

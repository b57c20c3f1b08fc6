import torch.nn.functional as F


def pad(img, padding, fill=0, padding_mode='constant'):
    l, t, r, b = padding
    mode = {'edge': 'replicate', 'constant': 'constant', 'reflect': 'reflect'}[padding_mode]
    x = img if img.dim() == 4 else img.unsqueeze(0)
    y = F.pad(x, (l, r, t, b), mode=mode) if mode != 'constant' else F.pad(x, (l, r, t, b), value=fill)
    return y if img.dim() == 4 else y.squeeze(0)


# torchvision 0.14 tensor-path photometric functions (third-party, absent here): the oracle's restatement stands in so that the reference's
# RCNet/rcnet_transforms.py can be imported and run by tests/golden/make_golden.py (fixture g13)
from oracle.transforms import adjust_brightness, adjust_contrast, adjust_saturation  # noqa: E402,F401

def to_pil_image(*a, **k):
    raise RuntimeError("torchvision stub: to_pil_image is not available (visualisation only)")


def to_tensor(*a, **k):
    raise RuntimeError("torchvision stub: to_tensor is not available (visualisation only)")

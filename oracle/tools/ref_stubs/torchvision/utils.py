def make_grid(*a, **k):
    raise RuntimeError("torchvision stub: make_grid is not available (visualisation only)")

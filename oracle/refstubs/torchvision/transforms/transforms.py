class Compose(object):
    """Published behaviour of torchvision.transforms.Compose: apply the callables in order."""

    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x

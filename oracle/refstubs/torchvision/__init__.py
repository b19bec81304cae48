"""
Stand-in for torchvision (absent from this image): the reference's evaluation helpers import
`torchvision.transforms.transforms.Compose` (reference eval/helpers.py:27,178-186) to chain the per-sample transforms of
its datasets.  Only that class exists here; used ONLY by tests/golden/make_golden.py (see ../README.md).
"""

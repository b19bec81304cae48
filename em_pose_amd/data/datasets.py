"""
Datasets over the on-disk formats of the path (reference empose/data/datasets.py:69-84 for the real recordings).  The
reference keeps AMASS / 3DPW in LMDB databases; the `lmdb` package and those databases are absent here, so the training
script reads the AMASS release's own `*.npz` sequences instead (`AMASSNpzDataset`).
"""
import glob
import os

from torch.utils.data import Dataset

from em_pose_amd.data.data import AMASSSample, RealSample


class RealDataset(Dataset):
    """The `*_clean.npz` recordings of one directory, optionally transformed."""

    def __init__(self, base_path, transform=None):
        self.files = sorted(glob.glob(os.path.join(base_path, '*_clean.npz')))
        self.transform = transform

    def __len__(self):
        return len(self.files)

    def __getitem__(self, item):
        sample = RealSample.from_npz_clean(self.files[item])
        return sample if self.transform is None else self.transform(sample)


class AMASSNpzDataset(Dataset):
    """AMASS sequences (`poses`, `betas`, `trans`, `mocap_framerate` per npz) found under `base_path`, recursively."""

    def __init__(self, base_path, transform=None, files=None):
        self.files = sorted(glob.glob(os.path.join(base_path, '**', '*.npz'), recursive=True)) if files is None else files
        self.transform = transform

    def __len__(self):
        return len(self.files)

    def __getitem__(self, item):
        path = self.files[item]
        sample = AMASSSample.from_disk(path, os.path.splitext(os.path.basename(path))[0])
        return sample if self.transform is None else self.transform(sample)

"""
Datasets over the on-disk formats of the path.

* `RealDataset`     -- the `*_clean.npz` recordings (reference empose/data/datasets.py:69-84).
* `LMDBDataset`     -- the key schema the reference keeps AMASS / 3DPW in (reader: reference datasets.py:42-59; writer:
                       scripts/preprocess_amass_3dpw.py:171-189).  Sequence i is seven records -- `poses{i}`, `betas{i}`,
                       `trans{i}`, `joints{i}` (raw float32 buffers, row-major, n_frames rows), `n_frames{i}`, `id{i}`,
                       `gender{i}` (utf-8 text) -- plus one `__len__` record.  The schema is read through ANY object
                       with `get(key: bytes) -> bytes | None`: an lmdb transaction when the `lmdb` package is
                       importable (it is not in this image), a dict, a shelve ... so the schema is tested without it.
* `AMASSNpzDataset` -- the AMASS release's own `*.npz` sequences, for a user without the LMDB conversion.
"""
import glob
import os

import numpy as np
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


# ---- LMDB key schema -------------------------------------------------------------------------------------------------
LMDB_ARRAY_FIELDS = ('poses', 'betas', 'trans', 'joints')   # float32 buffers
LMDB_TEXT_FIELDS = ('n_frames', 'id', 'gender')             # utf-8 text
LMDB_LEN_KEY = b'__len__'
LMDB_FPS = 60.0                                             # the conversion resamples everything to 60 Hz


def lmdb_key(field, index):
    return '{}{}'.format(field, index).encode()


def encode_sequence_records(index, seq_id, poses, betas, trans, joints, gender='unknown'):
    """The seven (key, value) records of sequence `index` as the reference's conversion script stores them
    (scripts/preprocess_amass_3dpw.py:171-178).  Used to build databases for tests and by users converting data."""
    n_frames = int(np.asarray(poses).shape[0])
    arrays = {'poses': poses, 'betas': betas, 'trans': trans, 'joints': joints}
    for name in ('trans', 'joints'):
        assert np.asarray(arrays[name]).shape[0] == n_frames, name
    rec = {lmdb_key(k, index): np.ascontiguousarray(v, dtype=np.float32).tobytes() for k, v in arrays.items()}
    rec[lmdb_key('n_frames', index)] = str(n_frames).encode()
    rec[lmdb_key('id', index)] = str(seq_id).encode()
    rec[lmdb_key('gender', index)] = (gender.decode() if isinstance(gender, bytes) else str(gender)).encode()
    return rec


def decode_sequence(get, index):
    """Sequence `index` out of a store, through `get(key) -> bytes` (reference datasets.py:42-59)."""
    def need(field):
        v = get(lmdb_key(field, index))
        if v is None:
            raise KeyError('record {!r} is missing'.format(lmdb_key(field, index).decode()))
        return bytes(v)
    n_frames = int(need('n_frames').decode())
    buf = {k: np.frombuffer(need(k), dtype=np.float32).copy() for k in LMDB_ARRAY_FIELDS}
    for k in ('poses', 'trans', 'joints'):
        if n_frames <= 0 or buf[k].size % n_frames:
            raise ValueError('{}{}: {} floats do not divide into {} frames'.format(k, index, buf[k].size, n_frames))
        buf[k] = buf[k].reshape(n_frames, -1)
    return AMASSSample(id=need('id').decode(), poses=buf['poses'], shape=buf['betas'], trans=buf['trans'],
                       fps=LMDB_FPS, joints=buf['joints'], gender=need('gender').decode())


class _LmdbFile(object):
    """`get` over a real LMDB file; the environment is opened lazily in the process that reads (DataLoader workers),
    with the reference's flags (datasets.py:33-37)."""

    def __init__(self, path):
        self.path, self.env = path, None

    def get(self, key):
        if self.env is None:
            try:
                import lmdb
            except ImportError as e:
                raise ImportError('reading {} needs the `lmdb` package (or pass any object with '
                                  'get(key: bytes) -> bytes instead of a path)'.format(self.path)) from e
            self.env = lmdb.open(self.path, subdir=os.path.isdir(self.path), readonly=True, lock=False,
                                 readahead=False, meminit=False)
        with self.env.begin(write=False) as txn:
            return txn.get(key)

    def close(self):
        if self.env is not None:
            self.env.close()
            self.env = None

    def __getstate__(self):      # never ship an open environment to a worker process
        return {'path': self.path, 'env': None}


class LMDBDataset(Dataset):
    """AMASS / 3DPW sequences stored under the reference's LMDB key schema.
    :param store: path of an LMDB database, or any object with `get(key: bytes) -> bytes | None`."""

    def __init__(self, store, transform=None):
        self.store = _LmdbFile(store) if isinstance(store, (str, bytes, os.PathLike)) else store
        self.transform = transform
        n = self.store.get(LMDB_LEN_KEY)
        if n is None:
            raise KeyError("the store has no '__len__' record")
        self.length = int(bytes(n).decode())
        if isinstance(self.store, _LmdbFile):
            self.store.close()   # reopened by whoever reads first (reference datasets.py:31-32)

    def __len__(self):
        return self.length

    def __getitem__(self, index):
        if not 0 <= index < self.length:
            raise IndexError(index)
        sample = decode_sequence(self.store.get, index)
        return sample if self.transform is None else self.transform(sample)

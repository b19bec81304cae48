"""
Build-authored stand-in for the `lmdb` package (absent from this image), used ONLY by tests/golden/make_golden.py so
that the UNMODIFIED reference reader (empose/data/datasets.py:19-60) can be run over a key-value store and its output
recorded.  Databases are plain dicts registered under a path name; only what that reader touches exists:
`open(path, **flags)` -> environment with `begin(write=False)` (a context manager yielding a transaction with
`get(key)` / `put(key, value)`) and `close()`.
"""
_DATABASES = {}


def register(path, records):
    _DATABASES[path] = records


class _Txn(object):
    def __init__(self, records, write):
        self._r, self._write = records, write

    def get(self, key):
        return self._r.get(bytes(key))

    def put(self, key, value):
        assert self._write
        self._r[bytes(key)] = bytes(value)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


class _Env(object):
    def __init__(self, records):
        self._r = records

    def begin(self, write=False):
        return _Txn(self._r, write)

    def close(self):
        pass


def open(path, **flags):
    return _Env(_DATABASES.setdefault(path, {}))

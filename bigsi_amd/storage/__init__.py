"""Storage registry, as in the reference (bigsi/storage/__init__.py:3-19): `storage-engine` names a backend class,
which is constructed from `storage-config`.  This package ships the one backend it is about -- `hip-hbm` -- and
third parties can register others in STORAGE_DICT exactly as the reference's backends do."""
from .contract import BaseStorage
from .hip_hbm import HipHbmStorage

STORAGE_DICT = {"hip-hbm": HipHbmStorage}


def get_storage(config):
    engine = config["storage-engine"]
    try:
        cls = STORAGE_DICT[engine]
    except KeyError:
        raise KeyError("storage-engine %r is not registered (available: %s); the berkeleydb/rocksdb/redis backends "
                       "belong to the reference package" % (engine, ", ".join(sorted(STORAGE_DICT))))
    storage_config = dict(config.get("storage-config") or {})
    if cls is HipHbmStorage:
        # the device matrix needs its row count before the first row arrives (BitMatrix.create stores rows first,
        # bigsi/matrix/bitmatrix.py:21-23): pass the top-level m / h down as hints
        for key in ("m", "h"):
            if key in config:
                storage_config.setdefault(key, config[key])
    return cls(storage_config)

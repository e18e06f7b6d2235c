"""Real-data input path: the reference's cached-sample LMDB (`<lmdb_dir>_cache`, written by
data_loader/data_preprocessor.py:308-333) read WITHOUT the `lmdb` package and WITHOUT `pyarrow.deserialize` (removed from
pyarrow; the reference pins pyarrow==11.0.0, requirements.txt:8), feeding device-side normalisation + DAE encoding."""
from .lmdb_format import LMDBReader, write_lmdb          # noqa: F401
from .arrow_legacy import deserialize, serialize        # noqa: F401
from .dataset import TrinityChunks, TrinityDataset_DAEed_Autoencoder        # noqa: F401

"""ctypes binding of the C oracle (oracle/libepik_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never by the product package `epik_amd`.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libepik_oracle.so")

ORC_ROW = np.dtype([("branch", np.uint32), ("score", np.float32), ("lwr", np.float64)])


class _OrcDb(ctypes.Structure):
    _fields_ = [
        ("kmer_size", ctypes.c_uint32),
        ("alphabet_size", ctypes.c_uint32),
        ("num_branches", ctypes.c_uint32),
        ("keep_at_most", ctypes.c_uint32),
        ("keep_factor", ctypes.c_double),
        ("threshold", ctypes.c_float),
        ("log_threshold", ctypes.c_float),
        ("num_keys", ctypes.c_uint64),
        ("offsets", ctypes.c_void_p),
        ("values", ctypes.c_void_p),
        ("char_class", ctypes.c_void_p),
        ("hash", ctypes.c_void_p),
    ]


def build(force: bool = False) -> str:
    """Compiles the oracle with the committed Makefile (gcc); returns the .so path."""
    src_newer = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
        for f in ("epik_oracle.c", "epik_oracle.h", "Makefile"))
    if force or src_newer:
        subprocess.run(["make", "-C", _HERE, "-B", "libepik_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        lib = ctypes.CDLL(_LIB_PATH)
        lib.orc_place_batch.restype = ctypes.c_int
        lib.orc_place_batch.argtypes = [ctypes.POINTER(_OrcDb), ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p,
                                        ctypes.c_void_p, ctypes.c_void_p]
        lib.orc_algorithmic_bytes.restype = ctypes.c_uint64
        lib.orc_algorithmic_bytes.argtypes = [ctypes.POINTER(_OrcDb), ctypes.c_void_p,
                                              ctypes.c_size_t, ctypes.c_uint32]
        lib.orc_max_threads.restype = ctypes.c_int
        lib.orc_place_batched.restype = ctypes.c_int
        lib.orc_place_batched.argtypes = [ctypes.POINTER(_OrcDb), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64,
                                          ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                          ctypes.c_void_p]
        lib.orc_hash_create.restype = ctypes.c_void_p
        lib.orc_hash_create.argtypes = [ctypes.POINTER(_OrcDb)]
        lib.orc_hash_create_sparse.restype = ctypes.c_void_p
        lib.orc_hash_create_sparse.argtypes = [ctypes.POINTER(_OrcDb), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]
        lib.orc_hash_destroy.restype = None
        lib.orc_hash_destroy.argtypes = [ctypes.c_void_p]
        _lib = lib
    return _lib


class Oracle:
    """Holds a CSR database + placer parameters and places read batches on the CPU."""

    def __init__(self, offsets, values, char_class, *, kmer_size, alphabet_size, num_branches,
                 threshold, log_threshold, keep_at_most=7, keep_factor=0.01):
        self._lib = _load()
        self.offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self.values = np.ascontiguousarray(values)
        assert self.values.dtype.itemsize == 8
        self.char_class = np.ascontiguousarray(char_class, dtype=np.uint32)
        assert self.char_class.shape == (256,)
        self.keep_at_most = int(keep_at_most)
        self._db = _OrcDb(
            kmer_size=int(kmer_size), alphabet_size=int(alphabet_size),
            num_branches=int(num_branches), keep_at_most=int(keep_at_most),
            keep_factor=float(keep_factor), threshold=float(threshold),
            log_threshold=float(log_threshold), num_keys=int(self.offsets.shape[0] - 1),
            offsets=self.offsets.ctypes.data, values=self.values.ctypes.data,
            char_class=self.char_class.ctypes.data, hash=None)
        self._hash = None

    @classmethod
    def from_synth(cls, db, states=None, keep_at_most=7, keep_factor=0.01):
        from epik_amd import alphabet  # tables only; not the product path
        if getattr(db, "keys", None) is not None:
            db = db.densified()  # the oracle's direct index: an offset per possible code
        return cls(db.offsets, db.values, alphabet.char_class_table(states or db.states),
                   kmer_size=db.kmer_size, alphabet_size=db.alphabet_size,
                   num_branches=db.num_branches, threshold=db.threshold,
                   log_threshold=db.log_threshold, keep_at_most=keep_at_most,
                   keep_factor=keep_factor)

    @classmethod
    def from_sparse(cls, db, states=None, keep_at_most=7, keep_factor=0.01):
        """From the sparse form of a database (db.keys[present] ascending, db.offsets[present + 1]): search() through
        the hash map alone, no offset per possible code (amino k = 7 would need 10 GB of them)."""
        from epik_amd import alphabet  # tables only; not the product path
        assert getattr(db, "keys", None) is not None
        self = cls(np.zeros(1, dtype=np.uint64), db.values, alphabet.char_class_table(states or db.states),
                   kmer_size=db.kmer_size, alphabet_size=db.alphabet_size, num_branches=db.num_branches,
                   threshold=db.threshold, log_threshold=db.log_threshold, keep_at_most=keep_at_most,
                   keep_factor=keep_factor)
        self._keys = np.ascontiguousarray(db.keys, dtype=np.uint32)
        self._sparse_offsets = np.ascontiguousarray(db.offsets, dtype=np.uint64)
        self._db.num_keys = int(db.alphabet_size) ** int(db.kmer_size)
        self._db.offsets = None  # (not read while the map is in place)
        self._hash = self._lib.orc_hash_create_sparse(ctypes.byref(self._db), self._keys.ctypes.data,
                                                      self._sparse_offsets.ctypes.data, int(self._keys.shape[0]))
        if not self._hash:
            raise MemoryError("orc_hash_create_sparse")
        self._db.hash = self._hash
        return self

    def place(self, seqs, seq_offsets, num_threads: int = 1):
        """Returns (rows[n, keep_at_most] ORC_ROW, n_rows[n] uint32, counts[n, keep_at_most] uint32)."""
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        seq_offsets = np.ascontiguousarray(seq_offsets, dtype=np.uint64)
        n = int(seq_offsets.shape[0] - 1)
        rows = np.zeros((n, self.keep_at_most), dtype=ORC_ROW)
        n_rows = np.zeros(n, dtype=np.uint32)
        counts = np.zeros((n, self.keep_at_most), dtype=np.uint32)
        rc = self._lib.orc_place_batch(ctypes.byref(self._db), seqs.ctypes.data,
                                       seq_offsets.ctypes.data, n, int(num_threads),
                                       rows.ctypes.data, n_rows.ctypes.data, counts.ctypes.data)
        if rc != 0:
            raise RuntimeError(f"orc_place_batch failed: {rc}")
        return rows, n_rows, counts

    def use_hash_map(self, enabled: bool = True) -> None:
        """phylo_kmer_db::search through a node-chained hash map (the reference's data structure shape)
        instead of the direct index; same results."""
        if enabled and not self._hash:
            self._hash = self._lib.orc_hash_create(ctypes.byref(self._db))
            if not self._hash:
                raise MemoryError("orc_hash_create")
        self._db.hash = self._hash if enabled else None

    def __del__(self):
        try:
            if getattr(self, "_hash", None):
                self._db.hash = None
                self._lib.orc_hash_destroy(self._hash)
                self._hash = None
        except Exception:
            pass

    def place_batched(self, seqs, seq_offsets, batch_size: int = 2000, num_threads: int = 1):
        """`place` as the reference's driver runs it: batches of `batch_size` reads, each de-duplicated by
        content and placed in an OpenMP dynamic loop (main.cpp:332-344, place.cpp:201-275)."""
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        seq_offsets = np.ascontiguousarray(seq_offsets, dtype=np.uint64)
        n = int(seq_offsets.shape[0] - 1)
        rows = np.zeros((n, self.keep_at_most), dtype=ORC_ROW)
        n_rows = np.zeros(n, dtype=np.uint32)
        counts = np.zeros((n, self.keep_at_most), dtype=np.uint32)
        rc = self._lib.orc_place_batched(ctypes.byref(self._db), seqs.ctypes.data, seq_offsets.ctypes.data, n,
                                         int(batch_size), int(num_threads), rows.ctypes.data, n_rows.ctypes.data,
                                         counts.ctypes.data)
        if rc != 0:
            raise RuntimeError(f"orc_place_batched failed: {rc}")
        return rows, n_rows, counts

    def algorithmic_bytes(self, seqs, seq_offsets, n_rows) -> int:
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        total = 0
        for i in range(len(seq_offsets) - 1):
            b, e = int(seq_offsets[i]), int(seq_offsets[i + 1])
            total += self._lib.orc_algorithmic_bytes(ctypes.byref(self._db), seqs.ctypes.data + b,
                                                     e - b, int(n_rows[i]))
        return total

    @staticmethod
    def max_threads() -> int:
        return _load().orc_max_threads()

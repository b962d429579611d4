"""ctypes view of oracle/build/libkmersets_oracle.so (test infrastructure).

The oracle is the CPU restatement of the reference's hot path; see the headers
under oracle/ for the reference file:line each function follows.  Only tests,
__graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "build", "libkmersets_oracle.so")


def build(force=False):
    if force or not os.path.exists(LIB_PATH) or any(
        os.path.getmtime(os.path.join(ORACLE_DIR, f)) > os.path.getmtime(LIB_PATH)
        for f in os.listdir(ORACLE_DIR)
        if f.endswith((".h", ".cc"))
    ):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return LIB_PATH


_lib = None

u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")


def lib():
    global _lib
    if _lib is not None:
        return _lib
    # KSH_ORACLE_LIB: another build of the same sources (bench.py's cpu_baseline leg loads the copy it has
    # just compiled with -O3 -march=native on the host it runs on)
    L = C.CDLL(os.environ.get("KSH_ORACLE_LIB") or build())
    vp, i, i64, u64, u32 = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_uint32
    sig = {
        "ko_kmer_from_string": (u64, [C.c_char_p, i]),
        "ko_kmer_to_string": (None, [u64, i, C.c_char_p]),
        "ko_complement": (u64, [u64, i]),
        "ko_canonical": (u64, [u64, i]),
        "ko_next": (u64, [u64, i, i]),
        "ko_prev": (u64, [u64, i, i]),
        "ko_complement_string": (None, [C.c_char_p, i64]),
        "ko_bucket_and_key": (None, [i, i, u64, C.POINTER(i64), C.POINTER(u64)]),
        "ko_kmer_from_bucket_and_key": (u64, [i, i, i64, u64]),
        "ko_canonical_many": (None, [u64p, i64, i, u64p]),
        "ko_complement_many": (None, [u64p, i64, i, u64p]),
        "ko_range_split": (None, [i64, i64, i64, i64p, i64p]),
        "ko_dsu_new": (vp, [i]),
        "ko_dsu_free": (None, [vp]),
        "ko_dsu_find": (i, [vp, i]),
        "ko_dsu_unite": (None, [vp, i, i]),
        "ko_dsu_is_same": (i, [vp, i, i]),
        "ko_dsu_unite_parallel": (None, [vp, i32p, i32p, i64, i]),
        "ko_set_new": (vp, [i, i, i]),
        "ko_set_free": (None, [vp]),
        "ko_set_copy": (vp, [vp]),
        "ko_set_add_kmers": (None, [vp, u64p, i64]),
        "ko_set_remove_kmers": (None, [vp, u64p, i64]),
        "ko_set_contains": (i, [vp, u64]),
        "ko_set_size": (i64, [vp]),
        "ko_set_hash": (u64, [vp]),
        "ko_set_clear": (None, [vp]),
        "ko_set_kmers": (None, [vp, u64p]),
        "ko_set_add_set": (None, [vp, vp]),
        "ko_set_sub_set": (None, [vp, vp]),
        "ko_set_intersection": (vp, [vp, vp]),
        "ko_set_diff": (i64, [vp, vp]),
        "ko_strings_new": (vp, [C.c_char_p, i64p, i64]),
        "ko_strings_free": (None, [vp]),
        "ko_strings_count": (i64, [vp]),
        "ko_strings_total": (i64, [vp]),
        "ko_strings_get": (None, [vp, C.c_char_p, i64p]),
        "ko_unitigs_canonical": (vp, [vp]),
        "ko_spss_canonical": (vp, [vp]),
        "ko_spss_from_unitigs": (vp, [vp, i]),
        "ko_spss_variant": (vp, [vp, i]),
        "ko_unitigs_directed": (vp, [vp]),
        "ko_set_from_spss": (vp, [vp, i, i, i, i]),
        "ko_svb_max_compressed_bytes": (i64, [u32]),
        "ko_svb_encode_0124": (i64, [u32p, u32, u8p]),
        "ko_svb_decode_0124": (i64, [u8p, u32p, u32]),
        "ko_compact_from_strings": (vp, [vp, i, i, i]),
        "ko_compact_from_set": (vp, [vp]),
        "ko_compact_from_set_variant": (vp, [vp, i, i]),
        "ko_compact_free": (None, [vp]),
        "ko_compact_to_set": (vp, [vp, i]),
        "ko_compact_to_strings": (vp, [vp]),
        "ko_compact_size": (i64, [vp]),
        "ko_compact_weight": (i64, [vp]),
        "ko_compact_n_strings": (i64, [vp]),
        "ko_compact_n_words": (i64, [vp]),
        "ko_compact_words": (None, [vp, u64p]),
        "ko_compact_lengths_compressed_size": (i64, [vp]),
        "ko_compact_lengths_compressed": (None, [vp, u8p]),
        "ko_compact_sampled": (None, [vp, i32p, i, i, i64p, C.c_void_p]),
        "ko_compact_dump": (i, [vp, C.c_char_p]),
        "ko_compact_load": (vp, [C.c_char_p, i, i, i]),
        "ko_counter_new": (vp, [i, i, i]),
        "ko_counter_free": (None, [vp]),
        "ko_counter_add": (None, [vp, u64, i]),
        "ko_counter_get": (i, [vp, u64]),
        "ko_counter_size": (i64, [vp]),
        "ko_counter_from_fasta": (i, [vp, C.c_char_p, i64, i]),
        "ko_counter_from_reads": (None, [vp, C.c_char_p, i64, i]),
        "ko_counter_to_set": (vp, [vp, i, i, C.POINTER(i64)]),
        "ko_kss_build": (vp, [C.POINTER(vp), i, i32p, i, i, i]),
        "ko_kss_build_mt": (vp, [C.POINTER(vp), i, i32p, i, i, i]),
        "ko_kss_free": (None, [vp]),
        "ko_kss_size": (i, [vp]),
        "ko_kss_get": (vp, [vp, i]),
        "ko_kss_node": (vp, [vp, i]),
        "ko_kss_n_iterations": (i, [vp]),
        "ko_kss_iterations": (None, [vp, i64p]),
        "ko_kss_n_checkpoints": (i, [vp]),
        "ko_kss_checkpoints": (None, [vp, i64p, f32p]),
        "ko_kss_initial_weights": (None, [vp, i64p]),
        "ko_kss_stat": (i64, [vp, i]),
        "ko_kss_meta": (i64, [vp, C.c_char_p, i64]),
        "ko_kss_dump": (i, [vp, C.c_char_p, C.c_char_p]),
        "ko_kss_load": (vp, [C.c_char_p, C.c_char_p, i, i, i, i]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


# --------------------------------------------------------------------------- helpers
def kmer(s):
    return int(lib().ko_kmer_from_string(s.encode(), len(s)))


def kmer_str(bits, k):
    buf = C.create_string_buffer(k + 1)
    lib().ko_kmer_to_string(int(bits), k, buf)
    return buf.value.decode()


class Strings:
    """Owning wrapper over a vector<string> handle."""

    def __init__(self, handle):
        self.h = handle

    @classmethod
    def from_list(cls, strings):
        chars = "".join(strings).encode()
        lens = np.array([len(s) for s in strings], dtype=np.int64)
        return cls(lib().ko_strings_new(chars, lens, len(strings)))

    def to_list(self):
        L = lib()
        n = L.ko_strings_count(self.h)
        total = L.ko_strings_total(self.h)
        buf = C.create_string_buffer(total + 1)
        lens = np.zeros(n, dtype=np.int64)
        L.ko_strings_get(self.h, buf, lens)
        raw = buf.raw[:total].decode()
        out, at = [], 0
        for ln in lens:
            out.append(raw[at:at + int(ln)])
            at += int(ln)
        return out

    def __del__(self):
        if getattr(self, "h", None):
            lib().ko_strings_free(self.h)
            self.h = None


class Set:
    """Owning wrapper over the oracle's KmerSet."""

    def __init__(self, k, n, key_bytes, handle=None):
        self.k, self.n, self.key_bytes = k, n, key_bytes
        self.h = handle if handle is not None else lib().ko_set_new(k, n, key_bytes)

    @classmethod
    def from_kmers(cls, k, n, key_bytes, kmers):
        s = cls(k, n, key_bytes)
        s.add(kmers)
        return s

    def _wrap(self, handle):
        return Set(self.k, self.n, self.key_bytes, handle)

    def add(self, kmers):
        a = np.ascontiguousarray(kmers, dtype=np.uint64)
        lib().ko_set_add_kmers(self.h, a, a.size)

    def remove(self, kmers):
        a = np.ascontiguousarray(kmers, dtype=np.uint64)
        lib().ko_set_remove_kmers(self.h, a, a.size)

    def contains(self, kmer_bits):
        return bool(lib().ko_set_contains(self.h, int(kmer_bits)))

    def size(self):
        return int(lib().ko_set_size(self.h))

    def hash(self):
        return int(lib().ko_set_hash(self.h))

    def kmers(self):
        out = np.zeros(self.size(), dtype=np.uint64)
        lib().ko_set_kmers(self.h, out)
        return out

    def copy(self):
        return self._wrap(lib().ko_set_copy(self.h))

    def add_set(self, other):
        lib().ko_set_add_set(self.h, other.h)
        return self

    def sub_set(self, other):
        lib().ko_set_sub_set(self.h, other.h)
        return self

    def intersection(self, other):
        return self._wrap(lib().ko_set_intersection(self.h, other.h))

    def diff(self, other):
        return int(lib().ko_set_diff(self.h, other.h))

    def equals(self, other):
        return self.diff(other) == 0

    def unitigs(self):
        return Strings(lib().ko_unitigs_canonical(self.h)).to_list()

    def spss(self):
        return Strings(lib().ko_spss_canonical(self.h)).to_list()

    def spss_slow(self):
        """GetSPSSCanonical(kmer_set, fast = false)."""
        return Strings(lib().ko_spss_variant(self.h, 1)).to_list()

    def unitigs_directed(self):
        """GetUnitigs: the non-canonical variant."""
        return Strings(lib().ko_unitigs_directed(self.h)).to_list()

    def spss_directed(self):
        """GetSPSS: the non-canonical variant."""
        return Strings(lib().ko_spss_variant(self.h, 2)).to_list()

    def compact(self, canonical=True, fast=True):
        """KmerSetCompact::FromKmerSet(set, canonical, fast, 1)."""
        return Compact(self.k, self.n, self.key_bytes,
                       lib().ko_compact_from_set_variant(self.h, int(canonical), int(fast)))

    @classmethod
    def from_spss(cls, strings, k, n, key_bytes, canonical=True):
        st = Strings.from_list(strings)
        return cls(k, n, key_bytes, lib().ko_set_from_spss(st.h, k, n, key_bytes, int(canonical)))

    def __del__(self):
        if getattr(self, "h", None):
            lib().ko_set_free(self.h)
            self.h = None


class Counter:
    """KmerCounter<K, N, KeyType, uint8_t> (lib/core/kmer_counter.h)."""

    def __init__(self, k, n, key_bytes):
        self.k, self.n, self.key_bytes = k, n, key_bytes
        self.h = lib().ko_counter_new(k, n, key_bytes)

    def add(self, kmer_bits, v):
        lib().ko_counter_add(self.h, int(kmer_bits), int(v))

    def get(self, kmer_bits):
        return lib().ko_counter_get(self.h, int(kmer_bits))

    def size(self):
        return lib().ko_counter_size(self.h)

    def from_fasta(self, text, canonical=True):
        """0 ok, 1 = odd number of lines, 2 = invalid FASTA file."""
        return lib().ko_counter_from_fasta(self.h, text, len(text), int(canonical))

    def from_reads(self, reads, canonical=True):
        text = "\n".join(reads).encode()
        if reads:
            text += b"\n"
        lib().ko_counter_from_reads(self.h, text, len(text), int(canonical))

    def to_set(self, cutoff):
        cut = C.c_int64()
        s = Set(self.k, self.n, self.key_bytes, lib().ko_counter_to_set(self.h, self.key_bytes, int(cutoff), C.byref(cut)))
        return s, cut.value

    def __del__(self):
        if getattr(self, "h", None):
            lib().ko_counter_free(self.h)
            self.h = None


class Compact:
    def __init__(self, k, n, key_bytes, handle):
        self.k, self.n, self.key_bytes, self.h = k, n, key_bytes, handle

    @classmethod
    def from_strings(cls, strings, k, n, key_bytes):
        st = Strings.from_list(strings)
        return cls(k, n, key_bytes, lib().ko_compact_from_strings(st.h, k, n, key_bytes))

    @classmethod
    def load(cls, path, k, n, key_bytes):
        h = lib().ko_compact_load(path.encode(), k, n, key_bytes)
        if not h:
            raise IOError("failed to open file")
        return cls(k, n, key_bytes, h)

    def dump(self, path):
        if lib().ko_compact_dump(self.h, path.encode()) != 0:
            raise IOError("failed to open file")

    def to_set(self, canonical=True):
        return Set(self.k, self.n, self.key_bytes, lib().ko_compact_to_set(self.h, int(canonical)))

    def strings(self):
        return Strings(lib().ko_compact_to_strings(self.h)).to_list()

    def size(self):
        return int(lib().ko_compact_size(self.h))

    def weight(self):
        return int(lib().ko_compact_weight(self.h))

    def n_strings(self):
        return int(lib().ko_compact_n_strings(self.h))

    def words(self):
        out = np.zeros(lib().ko_compact_n_words(self.h), dtype=np.uint64)
        lib().ko_compact_words(self.h, out)
        return out

    def lengths_compressed(self):
        out = np.zeros(lib().ko_compact_lengths_compressed_size(self.h), dtype=np.uint8)
        lib().ko_compact_lengths_compressed(self.h, out)
        return out

    def sampled(self, bucket_ids, canonical=True):
        ids = np.ascontiguousarray(bucket_ids, dtype=np.int32)
        offsets = np.zeros(ids.size + 1, dtype=np.int64)
        lib().ko_compact_sampled(self.h, ids, ids.size, int(canonical), offsets, None)
        keys = np.zeros(int(offsets[-1]), dtype=np.uint64)
        lib().ko_compact_sampled(self.h, ids, ids.size, int(canonical), offsets,
                                 keys.ctypes.data_as(C.c_void_p))
        return offsets, keys

    def __del__(self):
        if getattr(self, "h", None):
            lib().ko_compact_free(self.h)
            self.h = None


class KmerSetSet:
    def __init__(self, compacts=None, bucket_ids=None, canonical=True, max_iterations=-1,
                 handle=None, geom=None, n_workers=1):
        """n_workers > 1: the reference's multi-worker branches (oracle/ko_mt.h; canonical sets only).
        Node sets, merge sequence and N_proc equal the n_workers == 1 build; the SPSS strings may
        not (thread interleaving decides the greedy matching in the reference as well)."""
        L = lib()
        if handle is not None:
            self.h = handle
            self.k, self.n, self.key_bytes = geom
            return
        self.k, self.n, self.key_bytes = compacts[0].k, compacts[0].n, compacts[0].key_bytes
        arr = (C.c_void_p * len(compacts))(*[c.h for c in compacts])
        ids = np.ascontiguousarray(bucket_ids, dtype=np.int32)
        if n_workers > 1:
            assert canonical, "the multi-worker port covers canonical sets"
            self.h = L.ko_kss_build_mt(arr, len(compacts), ids, ids.size, max_iterations, n_workers)
        else:
            self.h = L.ko_kss_build(arr, len(compacts), ids, ids.size, int(canonical), max_iterations)

    @classmethod
    def load(cls, directory, ext, k, n, key_bytes, canonical=True):
        h = lib().ko_kss_load(directory.encode(), ext.encode(), k, n, key_bytes, int(canonical))
        if not h:
            raise IOError("failed to open file")
        return cls(handle=h, geom=(k, n, key_bytes))

    def dump(self, directory, ext):
        os.makedirs(directory, exist_ok=True)
        if lib().ko_kss_dump(self.h, directory.encode(), ext.encode()) != 0:
            raise IOError("failed to write files")

    def size(self):
        return int(lib().ko_kss_size(self.h))

    def get(self, i):
        return Set(self.k, self.n, self.key_bytes, lib().ko_kss_get(self.h, i))

    def node(self, i):
        return Compact(self.k, self.n, self.key_bytes, lib().ko_kss_node(self.h, i))

    def iterations(self):
        n = lib().ko_kss_n_iterations(self.h)
        out = np.zeros((max(n, 1), 5), dtype=np.int64)
        lib().ko_kss_iterations(self.h, out.reshape(-1))
        return out[:n]

    def checkpoints(self):
        n = lib().ko_kss_n_checkpoints(self.h)
        out = np.zeros((max(n, 1), 4), dtype=np.int64)
        imp = np.zeros(max(n, 1), dtype=np.float32)
        lib().ko_kss_checkpoints(self.h, out.reshape(-1), imp)
        return out[:n], imp[:n]

    def initial_weights(self, n0):
        out = np.zeros(max(n0 * (n0 - 1) // 2, 1), dtype=np.int64)
        lib().ko_kss_initial_weights(self.h, out)
        return out[: n0 * (n0 - 1) // 2]

    def stat(self, which):
        return int(lib().ko_kss_stat(self.h, which))

    def meta(self):
        n = lib().ko_kss_meta(self.h, None, 0)
        buf = C.create_string_buffer(n + 1)
        lib().ko_kss_meta(self.h, buf, n + 1)
        return buf.value.decode()

    def __del__(self):
        if getattr(self, "h", None):
            lib().ko_kss_free(self.h)
            self.h = None

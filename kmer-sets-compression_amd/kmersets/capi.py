"""ctypes binding of include/kmersets_hip.h (libkmersets_hip.so, HIP/gfx950).

Device memory is owned by torch tensors (plumbing); the C ABI only ever sees raw
device pointers.  There is no CPU fallback: if the library is missing or no GPU is
visible, calls raise.
"""
import ctypes as C
import os
import subprocess

import numpy as np

PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC_DIR = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(CSRC_DIR, "libkmersets_hip.so")
ROOT = os.path.dirname(PKG_DIR)
HEADER = os.path.join(ROOT, "include", "kmersets_hip.h")

KSH_OK, KSH_INVALID_ARGUMENT, KSH_FAILED_PRECONDITION, KSH_INTERNAL = 0, 3, 9, 13


class KshError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("ksh error %d: %s" % (code, message))
        self.code = code


class Geom(C.Structure):
    _fields_ = [("k", C.c_int32), ("n_bucket_bits", C.c_int32), ("key_bytes", C.c_int32),
                ("reserved", C.c_int32)]


class SetView(C.Structure):
    _fields_ = [("d_offsets", C.c_void_p), ("d_keys", C.c_void_p), ("n_keys", C.c_int64)]


class SpssView(C.Structure):
    _fields_ = [("d_words", C.c_void_p), ("d_lens", C.c_void_p), ("n_strings", C.c_int64),
                ("n_bases", C.c_int64)]


GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.c_int64, C.POINTER(C.c_int64))
COMM_ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
COMM_P2P_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32)
KSH_COMM_ID_BYTES = 128


COMM_ABORT_FN = C.CFUNCTYPE(None, C.c_void_p)


class CommFns(C.Structure):
    _fields_ = [("struct_size", C.c_size_t), ("user", C.c_void_p), ("allgather", COMM_ALLGATHER_FN), ("send", COMM_P2P_FN),
                ("recv", COMM_P2P_FN), ("abort", COMM_ABORT_FN)]


class PairJob(C.Structure):
    _fields_ = [("a", SetView), ("b", SetView), ("d_off_i", C.c_void_p), ("d_off_amb", C.c_void_p),
                ("d_off_bma", C.c_void_p), ("d_keys_i", C.c_void_p), ("d_keys_amb", C.c_void_p),
                ("d_keys_bma", C.c_void_p), ("totals", C.c_int64 * 3)]


def build(force=False):
    """Compiles the HIP sources for gfx950 (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC_DIR, f) for f in os.listdir(CSRC_DIR) if f.endswith((".hip", ".h"))]
    srcs.append(HEADER)
    if force or not os.path.exists(LIB_PATH) or any(
            os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs):
        subprocess.check_call(["make", "-C", CSRC_DIR, "-s", "-j4"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("KSH_LIB") or LIB_PATH   # KSH_LIB: another build of the same sources (tools/probe_trace.py)
    if not os.path.exists(path):
        raise RuntimeError(
            "%s is missing: run __graft_entry__.build() (make -C %s). "
            "The k-mer set hot path has no CPU fallback." % (path, CSRC_DIR))
    # torch ships its own libamdhip64; load it first so that this library binds to the
    # same HIP runtime (two runtimes in one process cannot both open the device).
    import torch  # noqa: F401

    L = C.CDLL(path)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    GP, SP = C.POINTER(Geom), C.POINTER(SetView)
    sig = {
        "ksh_version": (C.c_int, []),
        "ksh_last_error": (C.c_char_p, []),
        "ksh_device_count": (C.c_int, [C.POINTER(C.c_int)]),
        "ksh_malloc": (C.c_int, [C.c_int, C.c_size_t, C.POINTER(vp)]),
        "ksh_free": (C.c_int, [C.c_int, vp]),
        "ksh_memcpy_h2d": (C.c_int, [C.c_int, vp, vp, C.c_size_t]),
        "ksh_memcpy_d2h": (C.c_int, [C.c_int, vp, vp, C.c_size_t]),
        "ksh_memcpy_d2d": (C.c_int, [C.c_int, vp, vp, C.c_size_t]),
        "ksh_ctx_memcpy_h2d": (C.c_int, [vp, vp, vp, C.c_size_t]),
        "ksh_ctx_memcpy_d2h": (C.c_int, [vp, vp, vp, C.c_size_t]),
        "ksh_ctx_memcpy_d2d": (C.c_int, [vp, vp, vp, C.c_size_t]),
        "ksh_ctx_create": (C.c_int, [C.c_int, vp, C.POINTER(vp)]),
        "ksh_ctx_destroy": (C.c_int, [vp]),
        "ksh_ctx_sync": (C.c_int, [vp]),
        "ksh_ctx_reserve": (C.c_int, [vp, C.c_size_t]),
        "ksh_ctx_enable_timing": (C.c_int, [vp, C.c_int]),
        "ksh_ctx_timing_reset": (C.c_int, [vp]),
        "ksh_ctx_timing_read": (C.c_int, [vp, C.c_int, C.POINTER(C.c_float), C.POINTER(i64)]),
        "ksh_ctx_timing_units": (C.c_int, [vp, C.c_int, C.POINTER(i64)]),
        "ksh_ctx_set_lanes": (C.c_int, [vp, C.c_int]),
        "ksh_ctx_mem_stats": (C.c_int, [vp, C.POINTER(i64), C.c_int]),
        "ksh_ctx_timing_wall": (C.c_int, [vp, C.c_int, C.POINTER(C.c_float)]),
        "ksh_set_hash": (C.c_int, [vp, GP, SP, C.POINTER(C.c_uint64)]),
        "ksh_dsu_components": (C.c_int, [vp, i64, vp, vp, i64, vp]),
        "ksh_set_contains": (C.c_int, [vp, GP, SP, vp, i64, vp]),
        "ksh_set_kmers": (C.c_int, [vp, GP, SP, vp]),
        "ksh_pair_plan": (C.c_int, [vp, GP, SP, SP, vp, vp, vp, C.POINTER(i64)]),
        "ksh_pair_write": (C.c_int, [vp, GP, SP, SP, vp, vp, vp]),
        "ksh_set_diff": (C.c_int, [vp, GP, SP, SP, C.POINTER(i64)]),
        "ksh_pair_algebra": (C.c_int, [vp, GP, SP, SP, vp, vp, vp, vp, vp, vp, C.POINTER(i64)]),
        "ksh_pair_algebra_batch": (C.c_int, [vp, GP, C.POINTER(PairJob), i32]),
        "ksh_pair_weights": (C.c_int, [vp, GP, SP, i32, C.POINTER(i32), i32, C.POINTER(i32), i32,
                                       C.POINTER(i64)]),
        "ksh_spss_size": (C.c_int, [vp, GP, C.POINTER(SpssView), C.POINTER(i64)]),
        "ksh_spss_decode_plan": (C.c_int, [vp, GP, C.POINTER(SpssView), C.c_int, vp, C.POINTER(i64)]),
        "ksh_spss_decode_write": (C.c_int, [vp, GP, C.POINTER(SpssView), C.c_int, vp, vp,
                                            C.POINTER(i64)]),
        "ksh_fasta_plan": (C.c_int, [vp, GP, vp, i64, C.POINTER(i64), C.POINTER(i64)]),
        "ksh_fasta_write": (C.c_int, [vp, vp, vp]),
        "ksh_kmer_count_write": (C.c_int, [vp, GP, C.POINTER(SpssView), C.c_int, i32, vp, vp, C.POINTER(i64),
                                           C.POINTER(i64)]),
        "ksh_spss_to_text": (C.c_int, [vp, GP, C.POINTER(SpssView), vp]),
        "ksh_spss_from_text_plan": (C.c_int, [vp, GP, vp, i64, C.POINTER(i64), C.POINTER(i64)]),
        "ksh_spss_from_text_write": (C.c_int, [vp, vp, vp]),
        "ksh_spss_encode_plan": (C.c_int, [vp, GP, SP, C.c_int, C.c_int, C.POINTER(i64), C.POINTER(i64)]),
        "ksh_spss_encode_write": (C.c_int, [vp, vp, vp]),
        "ksh_spss_encode_stats": (C.c_int, [vp, C.POINTER(i64)]),
        "ksh_spss_encode_routes": (C.c_int, [vp, C.POINTER(i64)]),
        "ksh_spss_encode_release": (C.c_int, [vp]),
        "ksh_set_union_plan": (C.c_int, [vp, GP, SP, SP, vp, C.POINTER(i64)]),
        "ksh_set_union_write": (C.c_int, [vp, GP, SP, SP, vp]),
        "ksh_svb_encode_0124": (C.c_int, [vp, vp, i64, vp, C.POINTER(i64)]),
        "ksh_svb_decode_0124": (C.c_int, [vp, vp, i64, vp, C.POINTER(i64)]),
        "ksh_kss_build": (C.c_int, [vp, GP, C.POINTER(SpssView), i32, C.POINTER(i32), i32, C.c_int,
                                    i32, C.POINTER(vp)]),
        "ksh_kss_build_sharded": (C.c_int, [vp, GP, C.POINTER(SpssView), i32, C.POINTER(i32), i32, C.c_int,
                                            i32, i32, i32, GATHER_FN, vp, C.POINTER(vp)]),
        "ksh_kss_node_holder": (C.c_int, [vp, i32, C.POINTER(i32)]),
        "ksh_comm_unique_id": (C.c_int, [C.c_char_p]),
        "ksh_comm_create_rccl": (C.c_int, [vp, i32, i32, C.c_char_p, C.POINTER(vp)]),
        "ksh_comm_create_custom": (C.c_int, [vp, i32, i32, C.POINTER(CommFns), C.POINTER(vp)]),
        "ksh_comm_destroy": (C.c_int, [vp]),
        "ksh_kss_build_owned": (C.c_int, [vp, vp, GP, C.POINTER(SpssView), i32, C.POINTER(i32), C.POINTER(i32), i32,
                                          C.c_int, i32, C.POINTER(vp)]),
        "ksh_kss_comm_stats": (C.c_int, [vp, C.POINTER(i64)]),
        "ksh_comm_ranks_seen": (C.c_int, [vp, C.POINTER(i32)]),
        "ksh_kss_encode_counts": (C.c_int, [vp, C.POINTER(i64), C.POINTER(i64)]),
        "ksh_kss_weighed_counts": (C.c_int, [vp, C.POINTER(i64), C.POINTER(i64)]),
        "ksh_kss_phase_seconds": (C.c_int, [vp, C.POINTER(C.c_double)]),
        "ksh_kss_destroy": (C.c_int, [vp]),
        "ksh_kss_size": (C.c_int, [vp, C.POINTER(i32)]),
        "ksh_kss_node": (C.c_int, [vp, i32, C.POINTER(SpssView), SP, C.POINTER(i64)]),
        "ksh_kss_children": (C.c_int, [vp, i32, C.POINTER(C.POINTER(i32)), C.POINTER(i32)]),
        "ksh_kss_meta": (C.c_char_p, [vp]),
        "ksh_kss_trace": (C.c_int, [vp, C.POINTER(i64), C.POINTER(C.POINTER(i64)), C.POINTER(i64),
                                    C.POINTER(C.POINTER(i64)), C.POINTER(C.POINTER(C.c_float))]),
        "ksh_kss_initial_weights": (C.c_int, [vp, C.POINTER(C.POINTER(i64)), C.POINTER(i64)]),
        "ksh_kss_stats": (C.c_int, [vp, C.POINTER(i64)]),
        "ksh_kss_get": (C.c_int, [vp, i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(i64)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def exported_symbols():
    """Names declared in include/kmersets_hip.h (parsed from the header)."""
    import re

    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ksh_[a-z0-9_]+)\s*\(", text)))


def check(rc):
    if rc != KSH_OK:
        raise KshError(rc, lib().ksh_last_error().decode())


def geom(k, n_bucket_bits, key_bytes=None):
    """Device geometry of the reference's <K, N, KeyType>.  key_bytes: the reference's KeyType size (1, 2, 4
    or 8; None: the narrowest the key bits fit); device keys are 2, 4 or 8 bytes (uint8_t widens to 2)."""
    kb = 2 * k - n_bucket_bits
    if key_bytes is None:
        key_bytes = 2 if kb <= 16 else (4 if kb <= 32 else 8)
    return Geom(k, n_bucket_bits, 2 if key_bytes <= 2 else (4 if key_bytes <= 4 else 8), 0)


KEY_DTYPE = {2: np.uint16, 4: np.uint32, 8: np.uint64}


class DeviceSet:
    """A k-mer set resident in HBM: offsets int64[2^N+1] + sorted keys (u32/u64 bytes)."""

    def __init__(self, g, offsets, keys_bytes, n_keys):
        self.g, self.offsets, self.keys, self.n_keys = g, offsets, keys_bytes, int(n_keys)

    @classmethod
    def carved(cls, g, off_rows, row, key_pool, byte_start, byte_len):
        """A set whose offsets are row `row` of off_rows and whose keys are a byte range of
        key_pool; the tensor slices are only made when somebody asks for them (a batch of pair
        results carves dozens of these per call)."""
        self = cls.__new__(cls)
        self.g, self.n_keys = g, 0
        self._carve = (off_rows, row, key_pool, byte_start, byte_len)
        self._ptrs = (off_rows.data_ptr() + row * off_rows.shape[1] * 8, key_pool.data_ptr() + byte_start)
        return self

    def __getattr__(self, name):
        # only reached for attributes that are not set: the lazily sliced tensors of carved()
        if name in ("offsets", "keys") and "_carve" in self.__dict__:
            off_rows, row, key_pool, byte_start, byte_len = self._carve
            self.offsets = off_rows[row]
            self.keys = key_pool[byte_start:byte_start + byte_len]
            return self.__dict__[name]
        raise AttributeError(name)

    def pointers(self):
        p = self.__dict__.get("_ptrs")
        return p if p is not None else (self.offsets.data_ptr(), self.keys.data_ptr())

    @classmethod
    def from_numpy(cls, g, offsets, keys, device):
        import torch

        kdt = KEY_DTYPE[g.key_bytes]
        keys = np.ascontiguousarray(keys, dtype=kdt)
        off_t = torch.from_numpy(np.ascontiguousarray(offsets, dtype=np.int64)).to(device)
        raw = torch.from_numpy(keys.view(np.uint8).copy()) if keys.size else torch.zeros(0, dtype=torch.uint8)
        key_t = torch.empty(max(raw.numel(), 16), dtype=torch.uint8, device=device)
        key_t[: raw.numel()] = raw.to(device)
        return cls(g, off_t, key_t, keys.size)

    @classmethod
    def from_kmers(cls, g, kmers, device):
        from . import synth

        offsets, keys = synth.to_bucketed(kmers, g.k, g.n_bucket_bits, g.key_bytes)
        return cls.from_numpy(g, offsets, keys, device)

    @classmethod
    def empty_like_offsets(cls, g, n_keys, device):
        import torch

        off = torch.empty((1 << g.n_bucket_bits) + 1, dtype=torch.int64, device=device)
        keys = torch.empty(max(int(n_keys) * g.key_bytes, 16), dtype=torch.uint8, device=device)
        return cls(g, off, keys, n_keys)

    def view(self):
        po, pk = self.pointers()
        return SetView(po, pk, self.n_keys)

    def to_numpy(self):
        kdt = KEY_DTYPE[self.g.key_bytes]
        keys = self.keys[: self.n_keys * self.g.key_bytes].cpu().numpy().view(kdt)
        return self.offsets.cpu().numpy(), keys

    def kmers(self):
        from . import synth

        off, keys = self.to_numpy()
        return synth.from_bucketed(off, keys, self.g.k, self.g.n_bucket_bits)


class DeviceSpss:
    """KmerSetCompact on device: 2-bit bases in 64-bit words + (len - K) per string."""

    def __init__(self, g, words, lens, n_strings, n_bases):
        self.g, self.words, self.lens = g, words, lens
        self.n_strings, self.n_bases = int(n_strings), int(n_bases)

    @classmethod
    def from_strings(cls, g, strings, device):
        import torch
        from . import synth

        words, lens = synth.pack_strings(strings, g.k)
        w = torch.from_numpy(words.view(np.int64).copy()).to(device)
        ln = torch.from_numpy(lens.view(np.int32).copy()).to(device)
        if w.numel() == 0:
            w = torch.zeros(1, dtype=torch.int64, device=device)
        if ln.numel() == 0:
            ln = torch.zeros(1, dtype=torch.int32, device=device)
        return cls(g, w, ln, len(strings), int(sum(len(x) for x in strings)))

    def view(self):
        return SpssView(self.words.data_ptr(), self.lens.data_ptr(), self.n_strings, self.n_bases)

    def to_strings(self):
        from . import synth

        n_words = (self.n_bases + 31) // 32
        words = self.words[:n_words].cpu().numpy().view(np.uint64)
        lens = self.lens[: self.n_strings].cpu().numpy().view(np.uint32)
        return synth.unpack_strings(words, lens, self.g.k)

    def weight(self):
        return self.n_bases


class BatchResult:
    """Results of Context.pair_algebra_batch: indexable / iterable as [(A & B, A \\ B, B \\ A), ...];
    the DeviceSet objects (slices of the two batch-wide allocations) are made on first access;
    `totals` is the list [pair][|A&B|, |A\\B|, |B\\A|]."""

    def __init__(self, g, off_rows, key_pool, starts, byte_caps, totals):
        self.g, self.off_rows, self.key_pool, self.starts, self.byte_caps = g, off_rows, key_pool, starts, byte_caps
        self.totals = totals
        self._made = {}

    def __len__(self):
        return len(self.totals)

    def __getitem__(self, idx):
        if idx < 0:
            idx += len(self)
        if idx not in self._made:
            trio = []
            for r in range(3):
                s_ = DeviceSet.carved(self.g, self.off_rows, 3 * idx + r, self.key_pool,
                                      self.starts[3 * idx + r], self.byte_caps[3 * idx + r])
                s_.n_keys = self.totals[idx][r]
                trio.append(s_)
            self._made[idx] = trio
        return self._made[idx]

    def __iter__(self):
        return (self[i] for i in range(len(self)))


class Context:
    """ksh_ctx on one GPU; its stream becomes torch's current stream on that device."""

    def __init__(self, device_index=0):
        import torch

        if not torch.cuda.is_available():
            raise RuntimeError("no GPU visible: the k-mer set hot path has no CPU fallback")
        self.device = torch.device("cuda", device_index)
        torch.cuda.set_device(self.device)
        # One stream for everything: the library's kernels and torch's copies/allocations
        # are ordered against each other only if they share a stream (torch's default
        # stream has handle 0, which the C ABI reads as "make me a stream").
        self.stream = torch.cuda.Stream(device=self.device)
        torch.cuda.set_stream(self.stream)
        h = C.c_void_p()
        check(lib().ksh_ctx_create(device_index, C.c_void_p(self.stream.cuda_stream), C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            lib().ksh_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        # at interpreter shutdown the module globals may already be gone: the process's exit
        # releases the device memory then
        if lib is not None and _lib is not None:
            self.close()

    def sync(self):
        check(lib().ksh_ctx_sync(self.h))

    def enable_timing(self, on=True):
        check(lib().ksh_ctx_enable_timing(self.h, int(on)))

    def timing_reset(self):
        check(lib().ksh_ctx_timing_reset(self.h))

    def timing_read(self, kind):
        """(summed ms, launches) of the timed kernel kind since the last reset."""
        ms, n = C.c_float(), C.c_int64()
        check(lib().ksh_ctx_timing_read(self.h, kind, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def timing_units(self, kind):
        """k-mers (keys) covered by the timed launches of a kind since the last reset."""
        n = C.c_int64()
        check(lib().ksh_ctx_timing_units(self.h, kind, C.byref(n)))
        return n.value

    def timing_wall(self, kind):
        """ms during which at least one timed launch of the kind was running on any lane (union of the spans)."""
        ms = C.c_float()
        check(lib().ksh_ctx_timing_wall(self.h, kind, C.byref(ms)))
        return ms.value

    def mem_stats(self, reset_peak=False):
        """Device bytes behind the context: pooled buffers in use, their peak, cached, own scratch, lanes' total."""
        st = (C.c_int64 * 6)()
        check(lib().ksh_ctx_mem_stats(self.h, st, int(bool(reset_peak))))
        return {"pool_live": st[0], "pool_peak": st[1], "pool_cached": st[2], "scratch": st[3], "lanes": st[4],
                "n_lanes": st[5]}

    def set_lanes(self, n):
        """Independent jobs of one call (a check's encodes, the inputs' decodes) on up to n streams at once; 1: one
        stream, 0: the default (KSH_LANES, else 4)."""
        check(lib().ksh_ctx_set_lanes(self.h, int(n)))

    # KmerSet::Hash / Size -------------------------------------------------------
    def set_hash(self, s):
        out = C.c_uint64()
        v = s.view()
        check(lib().ksh_set_hash(self.h, C.byref(s.g), C.byref(v), C.byref(out)))
        return out.value

    # ParallelDisjointSet ----------------------------------------------------------------
    def dsu_components(self, n, xs, ys):
        """Unites the pairs (xs[i], ys[i]) over n nodes concurrently; returns Find(i) for every node."""
        import torch

        x = torch.from_numpy(np.ascontiguousarray(xs, dtype=np.int32)).to(self.device)
        y = torch.from_numpy(np.ascontiguousarray(ys, dtype=np.int32)).to(self.device)
        root = torch.empty(max(n, 1), dtype=torch.int32, device=self.device)
        check(lib().ksh_dsu_components(self.h, n, x.data_ptr() if x.numel() else None,
                                       y.data_ptr() if y.numel() else None, x.numel(), root.data_ptr()))
        return root[:n].cpu().numpy()

    # KmerSet::Contains / Find -----------------------------------------------------------
    def set_contains(self, s, kmers):
        """bool array: which of the 2K-bit patterns `kmers` are in the set (one launch)."""
        import torch

        q = torch.from_numpy(np.ascontiguousarray(kmers, dtype=np.uint64).view(np.int64).copy()).to(self.device)
        out = torch.zeros(max(q.numel(), 1), dtype=torch.uint8, device=self.device)
        v = s.view()
        check(lib().ksh_set_contains(self.h, C.byref(s.g), C.byref(v), q.data_ptr(), q.numel(), out.data_ptr()))
        return out[: q.numel()].cpu().numpy().astype(bool)

    def set_kmers(self, s):
        """All k-mers of the set as uint64 bit patterns, ascending (expanded on the device)."""
        import torch

        out = torch.empty(max(s.n_keys, 1), dtype=torch.int64, device=self.device)
        v = s.view()
        check(lib().ksh_set_kmers(self.h, C.byref(s.g), C.byref(v), out.data_ptr()))
        return out[: s.n_keys].cpu().numpy().view(np.uint64)

    # Intersection / Sub -----------------------------------------------------------
    def pair_plan(self, a, b, out_i, out_amb, out_bma):
        totals = (C.c_int64 * 3)()
        va, vb = a.view(), b.view()
        check(lib().ksh_pair_plan(self.h, C.byref(a.g), C.byref(va), C.byref(vb),
                                  out_i.offsets.data_ptr(), out_amb.offsets.data_ptr(),
                                  out_bma.offsets.data_ptr(), totals))
        return [int(x) for x in totals]

    def pair_write(self, a, b, out_i, out_amb, out_bma):
        va, vb = a.view(), b.view()
        check(lib().ksh_pair_write(self.h, C.byref(a.g), C.byref(va), C.byref(vb),
                                   out_i.keys.data_ptr() if out_i is not None else None,
                                   out_amb.keys.data_ptr() if out_amb is not None else None,
                                   out_bma.keys.data_ptr() if out_bma is not None else None))

    def pair_algebra(self, a, b):
        """(A & B, A \\ B, B \\ A) as new DeviceSets (exact-size key buffers)."""
        import torch

        g = a.g
        outs = [DeviceSet.empty_like_offsets(g, 0, self.device) for _ in range(3)]
        totals = self.pair_plan(a, b, *outs)
        for o, n in zip(outs, totals):
            o.n_keys = n
            o.keys = torch.empty(max(n * g.key_bytes, 16), dtype=torch.uint8, device=self.device)
        self.pair_write(a, b, *outs)
        return outs

    # KmerSetCompact::Size / ToKmerSet ---------------------------------------------------
    def spss_size(self, sp):
        out = C.c_int64()
        v = sp.view()
        check(lib().ksh_spss_size(self.h, C.byref(sp.g), C.byref(v), C.byref(out)))
        return out.value

    def spss_decode(self, sp, canonical=True):
        import torch

        g = sp.g
        out = DeviceSet.empty_like_offsets(g, 0, self.device)
        n = C.c_int64()
        v = sp.view()
        check(lib().ksh_spss_decode_plan(self.h, C.byref(g), C.byref(v), int(canonical),
                                         out.offsets.data_ptr(), C.byref(n)))
        out.keys = torch.empty(max(n.value * g.key_bytes, 16), dtype=torch.uint8, device=self.device)
        check(lib().ksh_spss_decode_write(self.h, C.byref(g), C.byref(v), int(canonical),
                                          out.offsets.data_ptr(), out.keys.data_ptr(), C.byref(n)))
        out.n_keys = n.value
        return out

    # KmerCounter::FromFASTA / FromReads / ToKmerSet -------------------------------------------
    def fasta_fragments(self, g, text):
        """uint8 tensor (device) holding FASTA text -> DeviceSpss of the reads' ACGT fragments
        of length >= K (what the counter counts k-mers of)."""
        import torch

        n_frag, n_bases = C.c_int64(), C.c_int64()
        check(lib().ksh_fasta_plan(self.h, C.byref(g), text.data_ptr() if text.numel() else None,
                                   text.numel(), C.byref(n_frag), C.byref(n_bases)))
        words = torch.zeros(max((n_bases.value + 31) // 32, 1), dtype=torch.int64, device=self.device)
        lens = torch.zeros(max(n_frag.value, 1), dtype=torch.int32, device=self.device)
        check(lib().ksh_fasta_write(self.h, words.data_ptr(), lens.data_ptr()))
        return DeviceSpss(g, words, lens, n_frag.value, n_bases.value)

    def kmer_count(self, reads, cutoff, canonical=True):
        """(KmerSet of the k-mers seen >= cutoff times in the fragments, number of distinct
        k-mers below the cutoff)."""
        import torch

        g = reads.g
        out = DeviceSet.empty_like_offsets(g, 0, self.device)
        n, n_cut = C.c_int64(), C.c_int64()
        v = reads.view()
        check(lib().ksh_spss_decode_plan(self.h, C.byref(g), C.byref(v), int(canonical),
                                         out.offsets.data_ptr(), C.byref(n)))
        out.keys = torch.empty(max(n.value * g.key_bytes, 16), dtype=torch.uint8, device=self.device)
        check(lib().ksh_kmer_count_write(self.h, C.byref(g), C.byref(v), int(canonical), int(cutoff),
                                         out.offsets.data_ptr(), out.keys.data_ptr(), C.byref(n),
                                         C.byref(n_cut)))
        out.n_keys = n.value
        return out, n_cut.value

    # KmerSetCompact::Dump / Load text (one string per line) ------------------------------
    def spss_to_text(self, sp):
        """The bytes KmerSetCompact::Dump writes, as a uint8 tensor on the device."""
        import torch

        text = torch.empty(sp.n_bases + sp.n_strings, dtype=torch.uint8, device=self.device)
        v = sp.view()
        check(lib().ksh_spss_to_text(self.h, C.byref(sp.g), C.byref(v), text.data_ptr() if text.numel() else None))
        return text

    def spss_from_text(self, g, text):
        """uint8 tensor (device) holding lines over ACGT -> DeviceSpss."""
        import torch

        n_strings, n_bases = C.c_int64(), C.c_int64()
        check(lib().ksh_spss_from_text_plan(self.h, C.byref(g), text.data_ptr() if text.numel() else None,
                                            text.numel(), C.byref(n_strings), C.byref(n_bases)))
        words = torch.empty(max((n_bases.value + 31) // 32, 1), dtype=torch.int64, device=self.device)
        lens = torch.empty(max(n_strings.value, 1), dtype=torch.int32, device=self.device)
        check(lib().ksh_spss_from_text_write(self.h, words.data_ptr(), lens.data_ptr()))
        return DeviceSpss(g, words, lens, n_strings.value, n_bases.value)

    # KmerSetCompact::FromKmerSet / GetUnitigsCanonical ---------------------------------------
    def spss_encode(self, s, mode=0, canonical=True):
        """mode 0: SPSS (GetSPSSCanonical fast, or GetSPSS when canonical is False); mode 1: unitigs;
        mode 2: GetSPSSCanonical(fast = false).  Returns a DeviceSpss."""
        import torch

        ns, nbases = C.c_int64(), C.c_int64()
        v = s.view()
        check(lib().ksh_spss_encode_plan(self.h, C.byref(s.g), C.byref(v), int(canonical), mode,
                                         C.byref(ns), C.byref(nbases)))
        n_words = (nbases.value + 31) // 32
        words = torch.empty(max(n_words, 1), dtype=torch.int64, device=self.device)
        lens = torch.empty(max(ns.value, 1), dtype=torch.int32, device=self.device)
        check(lib().ksh_spss_encode_write(self.h, words.data_ptr(), lens.data_ptr()))
        return DeviceSpss(s.g, words, lens, ns.value, nbases.value)

    def spss_encode_stats(self):
        st = (C.c_int64 * 4)()
        check(lib().ksh_spss_encode_stats(self.h, st))
        return {"unitigs": st[0], "rounds": st[1], "strings": st[2], "bases": st[3]}

    ROUTES = ("probe_staged", "rc_1024", "rc_512", "rc_256", "rc_64", "rc_batched", "scatter_two_level",
              "fwd_staged", "rank_one_launch", "heads_one_launch", "jump_two_level", "rank_stamped", "emit_logs",
              "long_stretches", "match_more_rounds", "fwd_targets", "rc1_streamed", "rc_marks_groups")

    def spss_encode_routes(self):
        """The kernel variants the last encode plan ran (KSH_ROUTE_* of include/kmersets_hip.h), as a set of names."""
        r = C.c_int64()
        check(lib().ksh_spss_encode_routes(self.h, C.byref(r)))
        return {name for bit, name in enumerate(self.ROUTES) if r.value >> bit & 1}

    def pair_algebra_onepass(self, a, b):
        """(A & B, A \\ B, B \\ A) by ksh_pair_algebra (count + write passes enqueued back to
        back); key buffers are allocated at their upper bounds before the sizes are known."""
        import torch

        g = a.g
        caps = (min(a.n_keys, b.n_keys), a.n_keys, b.n_keys)
        outs = []
        for cap in caps:
            o = DeviceSet.empty_like_offsets(g, 0, self.device)
            o.keys = torch.empty(max(cap * g.key_bytes, 16), dtype=torch.uint8, device=self.device)
            outs.append(o)
        totals = (C.c_int64 * 3)()
        va, vb = a.view(), b.view()
        check(lib().ksh_pair_algebra(self.h, C.byref(g), C.byref(va), C.byref(vb),
                                     outs[0].offsets.data_ptr(), outs[1].offsets.data_ptr(),
                                     outs[2].offsets.data_ptr(), outs[0].keys.data_ptr(),
                                     outs[1].keys.data_ptr(), outs[2].keys.data_ptr(), totals))
        for o, n in zip(outs, totals):
            o.n_keys = int(n)
        return outs

    def svb_encode(self, values):
        """StreamVByte-0124 bytes of a uint32 array (device round trip)."""
        import torch

        v = np.ascontiguousarray(values, dtype=np.uint32)
        d_in = torch.from_numpy(v.view(np.int32).copy()).to(self.device) if v.size else torch.zeros(1, dtype=torch.int32, device=self.device)
        cap = (v.size + 3) // 4 + 4 * v.size
        d_out = torch.empty(max(cap, 16), dtype=torch.uint8, device=self.device)
        n = C.c_int64()
        check(lib().ksh_svb_encode_0124(self.h, d_in.data_ptr(), v.size, d_out.data_ptr(), C.byref(n)))
        return d_out[: n.value].cpu().numpy()

    def svb_decode(self, data, n):
        import torch

        d = np.ascontiguousarray(data, dtype=np.uint8)
        d_in = torch.from_numpy(d.copy()).to(self.device) if d.size else torch.zeros(16, dtype=torch.uint8, device=self.device)
        d_out = torch.empty(max(n, 1), dtype=torch.int32, device=self.device)
        used = C.c_int64()
        check(lib().ksh_svb_decode_0124(self.h, d_in.data_ptr(), n, d_out.data_ptr(), C.byref(used)))
        return d_out[:n].cpu().numpy().view(np.uint32), used.value

    def pair_algebra_batch(self, pairs):
        """[(A, B), ...] -> [(A & B, A \\ B, B \\ A), ...]: every pair of the batch tiled together, one
        count launch, one write launch and one stream synchronisation for all sizes
        (ksh_pair_algebra_batch).  The result behaves like that list; `.totals` is the list
        [pair][|A&B|, |A\\B|, |B\\A|] for callers that only need the sizes."""
        import torch

        g = pairs[0][0].g
        kb = g.key_bytes
        n = len(pairs)
        # The job records as one int64 table with PairJob's layout (15 eight-byte fields):
        # a.{offsets, keys, n}, b.{offsets, keys, n}, 3 offset pointers, 3 key pointers, 3 totals.
        # Built with plain integers and converted once: a handful of pairs is far below the size
        # where numpy's per-call cost pays, and this runs once per merge step.
        assert C.sizeof(PairJob) == 15 * 8
        nb1 = (1 << g.n_bucket_bits) + 1
        byte_caps, starts = [], [0]
        for a, b in pairs:
            for cap in (min(a.n_keys, b.n_keys), a.n_keys, b.n_keys):
                nbytes = max(cap * kb, 16)
                byte_caps.append(nbytes)
                starts.append(starts[-1] + ((nbytes + 255) & ~255))       # bytes reserved per result
        # one offsets block and one key pool for the whole batch, carved by pointer arithmetic
        off_rows = torch.empty((3 * n, nb1), dtype=torch.int64, device=self.device)
        key_pool = torch.empty(starts[-1], dtype=torch.uint8, device=self.device)
        off0, key0 = off_rows.data_ptr(), key_pool.data_ptr()
        rows = []
        for idx, (a, b) in enumerate(pairs):
            pa, pb = a.pointers(), b.pointers()
            r = 3 * idx
            rows.append((pa[0], pa[1], a.n_keys, pb[0], pb[1], b.n_keys,
                         off0 + 8 * nb1 * r, off0 + 8 * nb1 * (r + 1), off0 + 8 * nb1 * (r + 2),
                         key0 + starts[r], key0 + starts[r + 1], key0 + starts[r + 2], 0, 0, 0))
        jobs = np.array(rows, dtype=np.int64)
        check(lib().ksh_pair_algebra_batch(self.h, C.byref(g), C.cast(jobs.ctypes.data, C.POINTER(PairJob)), n))
        return BatchResult(g, off_rows, key_pool, starts, byte_caps, jobs[:, 12:15].tolist())

    def set_union(self, a, b):
        """KmerSet::Add: A | B as a new DeviceSet."""
        import torch

        g = a.g
        out = DeviceSet.empty_like_offsets(g, 0, self.device)
        total = C.c_int64()
        va, vb = a.view(), b.view()
        check(lib().ksh_set_union_plan(self.h, C.byref(g), C.byref(va), C.byref(vb),
                                       out.offsets.data_ptr(), C.byref(total)))
        out.n_keys = total.value
        out.keys = torch.empty(max(total.value * g.key_bytes, 16), dtype=torch.uint8, device=self.device)
        check(lib().ksh_set_union_write(self.h, C.byref(g), C.byref(va), C.byref(vb),
                                        out.keys.data_ptr()))
        return out

    def set_diff(self, a, b):
        out = C.c_int64()
        va, vb = a.view(), b.view()
        check(lib().ksh_set_diff(self.h, C.byref(a.g), C.byref(va), C.byref(vb), C.byref(out)))
        return out.value

    def pair_weights(self, sets, bucket_ids, pairs):
        g = sets[0].g
        views = (SetView * len(sets))(*[s.view() for s in sets])
        ids = np.ascontiguousarray(bucket_ids, dtype=np.int32)
        prs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1)
        n_pairs = prs.size // 2
        out = np.zeros(max(n_pairs, 1), dtype=np.int64)
        check(lib().ksh_pair_weights(
            self.h, C.byref(g), views, len(sets),
            ids.ctypes.data_as(C.POINTER(C.c_int32)), ids.size,
            prs.ctypes.data_as(C.POINTER(C.c_int32)), n_pairs,
            out.ctypes.data_as(C.POINTER(C.c_int64))))
        return out[:n_pairs]


class DeviceKmerSetSet:
    """ksh_kss: KmerSetSet built on device (lib/core/kmer_set_set.h:109-427)."""

    def __init__(self, ctx, compacts, bucket_ids, canonical=True, max_iterations=-1, dist=None,
                 dist_device="cpu"):
        """dist: an initialised torch.distributed module (one process per GPU) for the sharded
        build (ksh_kss_build_sharded): every rank passes the same inputs; the SPSS of a node
        lives on one rank (node_holder)."""
        self.ctx, self.g, self.inputs = ctx, compacts[0].g, list(compacts)
        views = (SpssView * len(compacts))(*[c.view() for c in compacts])
        ids = np.ascontiguousarray(bucket_ids, dtype=np.int32)
        h = C.c_void_p()
        if dist is None:
            check(lib().ksh_kss_build(ctx.h, C.byref(self.g), views, len(compacts),
                                      ids.ctypes.data_as(C.POINTER(C.c_int32)), ids.size, int(canonical),
                                      max_iterations, C.byref(h)))
        else:
            import torch

            world, rank = dist.get_world_size(), dist.get_rank()

            def gather(_user, send, count, recv):
                try:
                    mine = torch.from_numpy(np.ctypeslib.as_array(send, shape=(count,)).copy()).to(dist_device)
                    parts = [torch.empty_like(mine) for _ in range(world)]
                    dist.all_gather(parts, mine)
                    out = np.ctypeslib.as_array(recv, shape=(world * count,))
                    for r, part in enumerate(parts):
                        out[r * count:(r + 1) * count] = part.cpu().numpy()
                    return 0
                except Exception:  # a Python exception must not unwind through the C frames
                    import traceback

                    traceback.print_exc()
                    return 1

            self._gather = GATHER_FN(gather)  # keep the callback object alive
            check(lib().ksh_kss_build_sharded(ctx.h, C.byref(self.g), views, len(compacts),
                                              ids.ctypes.data_as(C.POINTER(C.c_int32)), ids.size,
                                              int(canonical), max_iterations, rank, world, self._gather, None,
                                              C.byref(h)))
        self.h = h

    def node_holder(self, i):
        """Rank holding node i's SPSS in a sharded build; -1: every rank does."""
        r = C.c_int32()
        check(lib().ksh_kss_node_holder(self.h, i, C.byref(r)))
        return r.value

    def close(self):
        if getattr(self, "h", None):
            lib().ksh_kss_destroy(self.h)
            self.h = None

    __del__ = close

    def size(self):
        n = C.c_int32()
        check(lib().ksh_kss_size(self.h, C.byref(n)))
        return n.value

    def node_strings(self, i):
        """SPSS strings of node i (downloaded)."""
        from . import synth

        sv = SpssView()
        check(lib().ksh_kss_node(self.h, i, C.byref(sv), None, None))
        n_words = (sv.n_bases + 31) // 32
        words = np.zeros(max(n_words, 1), dtype=np.uint64)
        lens = np.zeros(max(sv.n_strings, 1), dtype=np.uint32)
        if n_words:
            check(lib().ksh_ctx_memcpy_d2h(self.ctx.h, words.ctypes.data_as(C.c_void_p), sv.d_words, n_words * 8))
        if sv.n_strings:
            check(lib().ksh_ctx_memcpy_d2h(self.ctx.h, lens.ctypes.data_as(C.c_void_p), sv.d_lens, sv.n_strings * 4))
        return synth.unpack_strings(words[:n_words], lens[: sv.n_strings], self.g.k)

    def node_size(self, i):
        n = C.c_int64()
        check(lib().ksh_kss_node(self.h, i, None, None, C.byref(n)))
        return n.value

    def node_kmers(self, i):
        from . import synth

        v = SetView()
        check(lib().ksh_kss_node(self.h, i, None, C.byref(v), None))
        return self._download_set(v.d_offsets, v.d_keys, v.n_keys)

    def _download_set(self, d_off, d_keys, n_keys):
        from . import synth

        g = self.g
        nb = 1 << g.n_bucket_bits
        off = np.zeros(nb + 1, dtype=np.int64)
        kdt = KEY_DTYPE[g.key_bytes]
        keys = np.zeros(max(n_keys, 1), dtype=kdt)
        check(lib().ksh_ctx_memcpy_d2h(self.ctx.h, off.ctypes.data_as(C.c_void_p), d_off, (nb + 1) * 8))
        if n_keys:
            check(lib().ksh_ctx_memcpy_d2h(self.ctx.h, keys.ctypes.data_as(C.c_void_p), d_keys, n_keys * g.key_bytes))
        return synth.from_bucketed(off, keys[:n_keys], g.k, g.n_bucket_bits)

    def children(self, i):
        p, n = C.POINTER(C.c_int32)(), C.c_int32()
        check(lib().ksh_kss_children(self.h, i, C.byref(p), C.byref(n)))
        return [p[j] for j in range(n.value)]

    def meta(self):
        return lib().ksh_kss_meta(self.h).decode()

    def trace(self):
        ni, nc = C.c_int64(), C.c_int64()
        rows, crows, imp = C.POINTER(C.c_int64)(), C.POINTER(C.c_int64)(), C.POINTER(C.c_float)()
        check(lib().ksh_kss_trace(self.h, C.byref(ni), C.byref(rows), C.byref(nc), C.byref(crows),
                                  C.byref(imp)))
        it = np.array([rows[i] for i in range(5 * ni.value)], dtype=np.int64).reshape(-1, 5)
        cp = np.array([crows[i] for i in range(4 * nc.value)], dtype=np.int64).reshape(-1, 4)
        im = np.array([imp[i] for i in range(nc.value)], dtype=np.float32)
        return it, cp, im

    def initial_weights(self):
        p, n = C.POINTER(C.c_int64)(), C.c_int64()
        check(lib().ksh_kss_initial_weights(self.h, C.byref(p), C.byref(n)))
        return np.array([p[i] for i in range(n.value)], dtype=np.int64)

    def stats(self):
        st = (C.c_int64 * 8)()
        check(lib().ksh_kss_stats(self.h, st))
        keys = ["initial_total_size", "final_total_size", "initial_spss_weight", "n_processed",
                "final_spss_weight", "packed_bytes", "length_bytes", "nodes"]
        out = dict(zip(keys, [int(x) for x in st]))
        ne, nk = C.c_int64(), C.c_int64()
        check(lib().ksh_kss_encode_counts(self.h, C.byref(ne), C.byref(nk)))
        out["n_encodes"], out["n_encoded_kmers"] = ne.value, nk.value
        check(lib().ksh_kss_weighed_counts(self.h, C.byref(ne), C.byref(nk)))
        out["n_weighed"], out["n_weighed_kmers"] = ne.value, nk.value
        ph = (C.c_double * 4)()
        check(lib().ksh_kss_phase_seconds(self.h, ph))
        out["phase_seconds"] = dict(zip(["decode_inputs", "weights", "merges", "encodes"], [float(x) for x in ph]))
        return out

    def get_size_and_hash(self, i):
        """(KmerSet::Size, KmerSet::Hash) of KmerSetSet::Get(i), computed on the device: what
        kmerset-multiple-compress --check compares per input set (src/kmerset-multiple-compress.cc:104-126)."""
        d_off, d_keys, n = C.c_void_p(), C.c_void_p(), C.c_int64()
        check(lib().ksh_kss_get(self.h, i, C.byref(d_off), C.byref(d_keys), C.byref(n)))
        try:
            v = SetView(d_off.value, d_keys.value, n.value)
            out = C.c_uint64()
            check(lib().ksh_set_hash(self.ctx.h, C.byref(self.g), C.byref(v), C.byref(out)))
            return n.value, out.value
        finally:
            dev = self.ctx.device.index
            lib().ksh_free(dev, d_off)
            lib().ksh_free(dev, d_keys)

    def get_kmers(self, i):
        """KmerSetSet::Get(i) as a sorted uint64 array (downloaded)."""
        d_off, d_keys, n = C.c_void_p(), C.c_void_p(), C.c_int64()
        check(lib().ksh_kss_get(self.h, i, C.byref(d_off), C.byref(d_keys), C.byref(n)))
        try:
            return self._download_set(d_off, d_keys, n.value)
        finally:
            dev = self.ctx.device.index
            lib().ksh_free(dev, d_off)
            lib().ksh_free(dev, d_keys)


class Comm:
    """ksh_comm: the transport of the owner-sharded build.  backend "nccl": RCCL inside the library on
    device buffers (rank 0 draws the id, torch.distributed carries its 128 bytes to the others);
    anything else: the caller-supplied functions, here gloo through host memory (rehearsals with
    several ranks on one GPU, where RCCL cannot run)."""

    def __init__(self, ctx, dist, coll_dev):
        import torch

        self.ctx, self.dist = ctx, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        h = C.c_void_p()
        if dist.get_backend() == "nccl":
            buf = C.create_string_buffer(KSH_COMM_ID_BYTES)
            if self.rank == 0:
                check(lib().ksh_comm_unique_id(buf))
            t = torch.tensor(list(buf.raw), dtype=torch.uint8, device=coll_dev)
            dist.broadcast(t, 0)
            raw = bytes(t.cpu().tolist())
            check(lib().ksh_comm_create_rccl(ctx.h, self.rank, self.world, raw, C.byref(h)))
            self.kind = "rccl"
        else:
            dev = ctx.device.index

            def d2h(ptr, n):
                a = np.empty(n, dtype=np.uint8)
                if n:
                    check(lib().ksh_ctx_memcpy_d2h(ctx.h, a.ctypes.data_as(C.c_void_p), ptr, n))
                return torch.from_numpy(a)

            def h2d(ptr, t):
                a = t.numpy()
                if a.size:
                    check(lib().ksh_ctx_memcpy_h2d(ctx.h, ptr, a.ctypes.data_as(C.c_void_p), a.size))

            def guarded(f):
                def g(*args):
                    try:
                        f(*args)
                        return 0
                    except Exception:  # a Python exception must not unwind through the C frames
                        import traceback

                        traceback.print_exc()
                        return 1
                return g

            @guarded
            def allgather(_u, d_send, d_recv, n):
                mine = d2h(d_send, n)
                parts = [torch.empty_like(mine) for _ in range(self.world)]
                dist.all_gather(parts, mine)
                h2d(d_recv, torch.cat(parts))

            @guarded
            def send(_u, d_buf, n, peer):
                dist.send(d2h(d_buf, n), peer)

            @guarded
            def recv(_u, d_buf, n, peer):
                t = torch.empty(n, dtype=torch.uint8)
                dist.recv(t, peer)
                h2d(d_buf, t)

            def abort(_u):
                # a rank gives the build up where it cannot follow the protocol: the transport's own abort, if any
                f = getattr(dist, "abort", None)
                if f is not None:
                    f()

            self._keep = (COMM_ALLGATHER_FN(allgather), COMM_P2P_FN(send), COMM_P2P_FN(recv), COMM_ABORT_FN(abort))
            self._fns = CommFns(C.sizeof(CommFns), None, *self._keep)
            check(lib().ksh_comm_create_custom(ctx.h, self.rank, self.world, C.byref(self._fns), C.byref(h)))
            self.kind = "custom"
        self.h = h

    def ranks_seen(self):
        """Collective: the ranks whose ids arrived here through the transport's all-gather."""
        n = C.c_int32()
        check(lib().ksh_comm_ranks_seen(self.h, C.byref(n)))
        return n.value

    def close(self):
        if getattr(self, "h", None):
            lib().ksh_comm_destroy(self.h)
            self.h = None

    __del__ = close


def block_owners(n_sets, world):
    """Input i lives on rank i * world // n_sets: contiguous blocks (neighbours in the input order are
    the likeliest to merge first)."""
    return [i * world // n_sets for i in range(n_sets)]


class OwnedKmerSetSet(DeviceKmerSetSet):
    """ksh_kss_build_owned: the multi-GPU KmerSetSet in which a node's set and SPSS live on one rank.
    compacts[i] is the input's DeviceSpss on its owner and None elsewhere."""

    def __init__(self, ctx, compacts, bucket_ids, dist, coll_dev, canonical=True, max_iterations=-1, owners=None,
                 comm=None):
        self.ctx, self.dist, self.coll_dev = ctx, dist, coll_dev
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.owners = list(owners) if owners is not None else block_owners(len(compacts), self.world)
        mine = [c for c in compacts if c is not None]
        if not mine:
            raise ValueError("rank %d owns no input" % self.rank)
        self.g, self.inputs = mine[0].g, list(compacts)
        for i, c in enumerate(compacts):
            if (c is not None) != (self.owners[i] == self.rank):
                raise ValueError("input %d: the container must be given on its owner (rank %d) only" % (i, self.owners[i]))
        views = (SpssView * len(compacts))(*[c.view() if c is not None else SpssView(None, None, 0, 0) for c in compacts])
        own = np.ascontiguousarray(self.owners, dtype=np.int32)
        ids = np.ascontiguousarray(bucket_ids, dtype=np.int32)
        self._own_comm = comm is None
        self.comm = comm if comm is not None else Comm(ctx, dist, coll_dev)
        h = C.c_void_p()
        check(lib().ksh_kss_build_owned(ctx.h, self.comm.h, C.byref(self.g), views, len(compacts),
                                        own.ctypes.data_as(C.POINTER(C.c_int32)),
                                        ids.ctypes.data_as(C.POINTER(C.c_int32)), ids.size, int(canonical),
                                        max_iterations, C.byref(h)))
        self.h = h
        self._table = None

    def close(self):
        DeviceKmerSetSet.close(self)
        if getattr(self, "_own_comm", False) and getattr(self, "comm", None) is not None:
            self.comm.close()
            self.comm = None

    __del__ = close

    def comm_stats(self):
        st = (C.c_int64 * 8)()
        check(lib().ksh_kss_comm_stats(self.h, st))
        return dict(zip(["p2p_bytes_sent", "p2p_bytes_received", "p2p_sets", "gather_bytes", "checks_deferred",
                         "rollbacks", "sets_migrated", "weight_gathers"], [int(x) for x in st]))

    def node_table(self):
        """(size, XOR hash) of every node, from the ranks that hold them (one all-gather)."""
        import torch

        if self._table is None:
            n = self.size()
            mine = np.zeros((n, 2), dtype=np.int64)
            for i in range(n):
                if self.node_holder(i) != self.rank:
                    continue
                v = SetView()
                check(lib().ksh_kss_node(self.h, i, None, C.byref(v), None))
                hsh = C.c_uint64()
                check(lib().ksh_set_hash(self.ctx.h, C.byref(self.g), C.byref(v), C.byref(hsh)))
                mine[i] = (v.n_keys, np.uint64(hsh.value).astype(np.int64))
            t = torch.from_numpy(mine).to(self.coll_dev)
            self.dist.all_reduce(t)          # every node is reported by exactly one rank
            self._table = t.cpu().numpy()
        return self._table

    def get_size_and_hash(self, i):
        """(Size, Hash) of Get(i): the nodes reachable from i are pairwise disjoint (every merge splits a
        set into its remainder and the new child), so the size is their sum and the hash their XOR."""
        tab = self.node_table()
        seen, todo = set(), [i]
        while todo:
            cur = todo.pop()
            if cur in seen:
                continue
            seen.add(cur)
            todo.extend(self.children(cur))
        size, hsh = 0, 0
        for node in seen:
            size += int(tab[node, 0])
            hsh ^= int(np.int64(tab[node, 1]).astype(np.uint64))
        return size, hsh

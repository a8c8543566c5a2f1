"""ctypes binding of oracle/liboracle.so -- the CPU checker (tests / bench cpu_baseline only)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_LIB = None


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(ORACLE_DIR, "scalce_oracle.c")):
        build()
    L = C.CDLL(so)
    vp, i32, i64, u64, sz = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_size_t
    L.orc_trie_from_bin.restype = vp
    L.orc_trie_from_bin.argtypes = [vp, sz]
    L.orc_trie_from_text.restype = vp
    L.orc_trie_from_text.argtypes = [C.c_char_p, sz]
    L.orc_trie_free.argtypes = [vp]
    for f in ("orc_trie_patterns", "orc_trie_nodes"):
        getattr(L, f).argtypes = [vp]
    L.orc_trie_pattern_len.argtypes = [vp, i32]
    L.orc_trie_pattern.argtypes = [vp, i32]
    L.orc_trie_pattern.restype = C.c_char_p
    L.orc_trie_pattern_id.argtypes = [vp, i32]
    L.orc_trie_reset_counts.argtypes = [vp]
    L.orc_tokenize_seq.argtypes = [vp, vp, i64, i32, i32, vp, vp]
    L.orc_bucket_order.argtypes = [vp, vp, i64, i32, i32, vp, vp, vp, vp]
    L.orc_pack_read.argtypes = [vp, i32, i32, i32, vp]
    L.orc_qmap_init.argtypes = [vp, vp, i32]
    L.orc_quality.argtypes = [vp, vp, i32, vp, vp, vp, vp, i32]
    L.orc_ac_scale.argtypes = [vp, i32, vp]
    L.orc_acstat_init.argtypes = [vp, vp]
    L.orc_ac_encode_block.restype = sz
    L.orc_ac_encode_block.argtypes = [vp, vp, sz, vp]
    L.orc_ac_decode_block.argtypes = [vp, vp, sz, vp]
    L.orc_ac_encode_stream.restype = sz
    L.orc_ac_encode_stream.argtypes = [vp, vp, sz, vp, sz, i32]
    _LIB = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class QMap(C.Structure):
    _fields_ = [("offset", C.c_int), ("values", C.c_int * 128)]


class Trie:
    def __init__(self, blob=None, text=None):
        L = lib()
        if blob is not None:
            buf = np.frombuffer(blob, dtype=np.uint8)
            self.h = L.orc_trie_from_bin(_p(buf), len(blob))
        else:
            self.h = L.orc_trie_from_text(text, len(text))
        if not self.h:
            raise ValueError("bad core table")
        self.n_patterns = L.orc_trie_patterns(self.h)
        self.n_nodes = L.orc_trie_nodes(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_trie_free(self.h)
            self.h = None

    def pattern(self, p):
        return lib().orc_trie_pattern(self.h, p)

    def pattern_lens(self):
        return np.array([lib().orc_trie_pattern_len(self.h, p) for p in range(self.n_patterns)], dtype=np.int32)

    def pattern_ids(self):
        return np.array([lib().orc_trie_pattern_id(self.h, p) for p in range(self.n_patterns)], dtype=np.int32)

    def tokenize(self, bases):
        """bases: (N, L) uint8 ASCII.  Sequential -T 1 semantics; counts start at zero."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        n, L = bases.shape
        pat = np.empty(n, dtype=np.int32)
        end = np.empty(n, dtype=np.int32)
        lib().orc_trie_reset_counts(self.h)
        lib().orc_tokenize_seq(self.h, _p(bases), n, L, L, _p(pat), _p(end))
        return pat, end

    def order(self, bases, pat, end, chunk=None):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        n, L = bases.shape
        perm = np.empty(n, dtype=np.int64)
        pat = np.ascontiguousarray(pat, dtype=np.int32)
        end = np.ascontiguousarray(end, dtype=np.int32)
        ch = None if chunk is None else np.ascontiguousarray(chunk, dtype=np.int32)
        lib().orc_bucket_order(self.h, _p(bases), n, L, L, _p(pat), _p(end), None if ch is None else _p(ch), _p(perm))
        return perm


def pack_read(row, n, l):
    row = np.ascontiguousarray(row, dtype=np.uint8)
    out = np.zeros(len(row) // 4 + 2, dtype=np.uint8)
    k = lib().orc_pack_read(_p(row), len(row), n, l, _p(out))
    return out[:k].copy()


def qmap_init(stat, lossy):
    q = QMap()
    st = np.ascontiguousarray(stat, dtype=np.int32)
    lib().orc_qmap_init(C.byref(q), _p(st), lossy)
    return q.offset, np.array(list(q.values), dtype=np.int32)


def quality_stream(quals, bases, offset, values, no_ac=False):
    """Apply output_quality to every read in input order.  Returns (q' (N,L) uint8, freq4 u64[512000])."""
    quals = np.ascontiguousarray(quals, dtype=np.uint8)
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    n, L = quals.shape
    q = QMap()
    q.offset = int(offset)
    for i in range(128):
        q.values[i] = int(values[i])
    out = np.empty((n, L), dtype=np.uint8)
    freq4 = np.zeros(512000, dtype=np.uint64)
    state = np.array([500, 500], dtype=np.uint32)
    f = lib().orc_quality
    qb, bb, ob = quals.ctypes.data, bases.ctypes.data, out.ctypes.data
    for r in range(n):
        f(qb + r * L, bb + r * L, L, C.byref(q), ob + r * L, _p(freq4), _p(state), int(no_ac))
    return out, freq4


def ac_scale(freq4, factor):
    out = np.empty(512000, dtype=np.uint32)
    f4 = np.ascontiguousarray(freq4, dtype=np.uint64)
    lib().orc_ac_scale(_p(f4), int(factor), _p(out))
    return out


class AcStat:
    SIZE = 80 * 80 * 80 * 4 * 2 + 80 * 80 * 4 + 8

    def __init__(self, table_u32):
        self.table = np.ascontiguousarray(table_u32, dtype=np.uint32)
        self.buf = np.zeros(self.SIZE + 64, dtype=np.uint8)
        lib().orc_acstat_init(_p(self.buf), _p(self.table))

    def encode_block(self, sym):
        sym = np.ascontiguousarray(sym, dtype=np.uint8)
        out = np.zeros(len(sym) * 2 + 64, dtype=np.uint8)
        k = lib().orc_ac_encode_block(_p(self.buf), _p(sym), len(sym), _p(out))
        return out[:k].copy()

    def decode_block(self, enc, nsym):
        enc = np.concatenate([np.ascontiguousarray(enc, dtype=np.uint8), np.zeros(8, dtype=np.uint8)])
        out = np.empty(nsym, dtype=np.uint8)
        lib().orc_ac_decode_block(_p(self.buf), _p(enc), nsym, _p(out))
        return out

    def encode_stream(self, sym, threads=1):
        sym = np.ascontiguousarray(sym, dtype=np.uint8)
        cap = len(sym) * 2 + 4096
        out = np.zeros(cap, dtype=np.uint8)
        k = lib().orc_ac_encode_stream(_p(self.buf), _p(sym), len(sym), _p(out), cap, threads)
        return out[:k].copy()


def orc_cli(*args, check=True):
    exe = os.path.join(ORACLE_DIR, "orc_cli")
    if not os.path.exists(exe):
        build()
    return subprocess.run([exe, *map(str, args)], check=check, capture_output=True)

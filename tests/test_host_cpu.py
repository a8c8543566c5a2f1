"""CPU (no GPU): the C-ABI library loads, exports what include/scalce_hip.h declares, refuses to run
without a device, and its host-side pieces (table builder, quality model) agree with the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oraclelib as O
from scalce_amd import host, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "scalce_hip.h")).read()
    names = sorted(set(re.findall(r"\b(scalce_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 30
    L = host.lib()
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_no_device_is_a_loud_error():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(host.ScalceError, match="no CPU path"):
        host.Context(0)


def _describe(blob, is_text):
    L = host.lib()
    L.scalce_patterns_describe_host.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t,
                                                C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    ns, nb = C.c_int32(), C.c_int32()
    assert L.scalce_patterns_describe_host(blob, len(blob), is_text, None, 0, C.byref(ns), C.byref(nb)) == 0
    out = np.zeros(nb.value + 1, dtype=np.int32)
    L.scalce_patterns_describe_host(blob, len(blob), is_text, out.ctypes.data, len(out), C.byref(ns), C.byref(nb))
    return ns.value, nb.value, out


def test_table_builder_matches_reference_bfs_order(patterns_blob):
    """Bucket emission order of the product's DFA builder == BFS ids of prepare_aho_automata (pinned by the
    reference objects through tests/golden/*.npz `ids`)."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "se100.npz"))
    ns, nb, order = _describe(patterns_blob, 0)
    ids = g["ids"]  # BFS id per pattern (file order)
    assert nb == 15600 and order[-1] == host.ROOT_CORE
    assert (order[:-1] == np.argsort(ids, kind="stable")).all()
    trie = O.Trie(blob=patterns_blob)
    assert ns == trie.n_nodes + 1
    # text list with nested and duplicated cores
    g2 = np.load(os.path.join(ROOT, "tests", "golden", "se150_text.npz"))
    txt = str(g2["ptxt"]).encode()
    ns2, nb2, order2 = _describe(txt, 1)
    ids2 = g2["ids"]
    live = np.flatnonzero(ids2 >= 0)  # a duplicated core keeps only its last index (reads.cpp:264)
    assert nb2 == len(live)
    assert (order2[:-1] == live[np.argsort(ids2[live], kind="stable")]).all()


def test_table_builder_rejects_garbage():
    L = host.lib()
    L.scalce_patterns_describe_host.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t,
                                                C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    assert L.scalce_patterns_describe_host(b"\x08\x00\xff\xff\xff\x7f", 6, 0, None, 0, None, None) != 0
    assert L.scalce_patterns_describe_host(b"\x40\x00\x01\x00\x00\x00" + b"\x00" * 16, 22, 0, None, 0, None, None) != 0


@pytest.mark.parametrize("lossy", [0, 10, 30, 60, 100])
def test_quality_model_matches_oracle(lossy):
    rng = np.random.default_rng(5 + lossy)
    for shape in ("normal", "bimodal", "phred64", "flat"):
        if shape == "normal":
            q = np.clip(np.rint(rng.normal(30, 8, 200000)), 2, 40).astype(int) + 33
        elif shape == "bimodal":
            q = np.concatenate([np.clip(np.rint(rng.normal(12, 3, 80000)), 2, 41),
                                np.clip(np.rint(rng.normal(36, 2, 120000)), 2, 41)]).astype(int) + 33
        elif shape == "phred64":
            q = np.clip(np.rint(rng.normal(28, 9, 200000)), 2, 40).astype(int) + 64
        else:
            q = rng.integers(35, 75, 200000)
        stat = np.bincount(q, minlength=128)[:128]
        off_o, vals_o = O.qmap_init(stat, lossy)
        off_h, vals_h = host.qmap_init(stat, lossy)
        assert off_o == off_h and (vals_o == vals_h).all(), (shape, lossy)


def test_sample_qmap_reads_like_the_reference():
    from scalce_amd import format as fmt
    b, q = synth.reads_and_quals(3000, 50, seed=9)
    fq = synth.fastq_bytes_fast(b, q)
    off, vals, L = fmt.sample_qmap(fq, sample=1000, lossy=30)
    stat = np.bincount(q[:1000].reshape(-1), minlength=128)
    off_o, vals_o = O.qmap_init(stat, 30)
    assert L == 50 and off == off_o and (vals == vals_o).all()


def test_renorm_count_formula_matches_the_literal_loop():
    """ac_encode_k's systolic step counts the renormalisation shifts of arithmetic.cpp:133-152 as
    clz((nlo mod 2^r) + D), D = nhi - nlo, r = top bit of D (kernels_ac.hpp: renorm_count).  Check the identity
    against the literal loop on random and crafted intervals; the only permitted disagreement is where the
    range renormalises to the full 2^32, which the kernel sends to the general path."""
    rng = np.random.default_rng(7)

    def literal(lo, hi):
        k = u = 0
        while k + u <= 40:
            if (hi >> 31) == (lo >> 31):
                k += 1
            elif (lo >> 30) & 1 and not (hi >> 30) & 1:
                u += 1
                lo &= 0x3FFFFFFF
                hi |= 0x40000000
            else:
                break
            lo = (lo << 1) & 0xFFFFFFFF
            hi = ((hi << 1) | 1) & 0xFFFFFFFF
        return k, u

    def clz(x):
        return 32 - int(x).bit_length()

    checked = 0
    for it in range(60000):
        mode = it % 6
        if mode == 0:
            a, b = sorted(int(x) for x in rng.integers(0, 2**32, 2))
        elif mode == 1:
            a = int(rng.integers(0, 2**32)); b = a + int(rng.integers(1, 1 << int(rng.integers(1, 32))))
        elif mode == 2:
            mid = 1 << int(rng.integers(1, 32)); base = int(rng.integers(0, 2**32)) // mid * mid
            a = base - int(rng.integers(1, 1 << int(rng.integers(1, 20)))); b = base + int(rng.integers(0, 1 << int(rng.integers(1, 20))))
        elif mode == 3:
            a = 0x3FFFFFFF - int(rng.integers(0, 1 << 16)); b = 0xC0000000 + int(rng.integers(0, 1 << 16))
        elif mode == 4:
            p = int(rng.integers(1, 31)); a = int(rng.integers(0, 2**32)) >> p << p; b = a + (1 << p) - 1
        else:
            a = int(rng.integers(0, 2**31)); b = int(rng.integers(2**31, 2**32))
        a = max(0, min(a, 2**32 - 1)); b = max(0, min(b, 2**32 - 1))
        if not a < b:
            continue
        k, u = literal(a, b)
        if k == 32:
            continue
        D = b - a
        width = (~clz(D)) & 31                      # v_bfe_u32 reads 5 bits of the width operand
        t = clz(((a & ((1 << width) - 1)) + D) & 0xFFFFFFFF)
        full = (((D + 1) << (k + u)) & 0xFFFFFFFF) == 0
        assert t == k + u or full, (hex(a), hex(b), k, u, t)
        checked += 1
    assert checked > 50000


def test_fastq_text_bytes_matches_the_text_it_sizes():
    """scalce_fastq_text_bytes sizes the device buffer scalce_fastq_records fills: with names it is the name stream
    minus the length bytes plus 2L + 6 per record ('@', two newlines around the bases, "+\\n", the final newline); in
    library mode the decimal index of every record is counted in closed form (decompress.cpp:290-304)."""
    L = host.lib()
    rng = np.random.default_rng(5)
    for n, rl in ((0, 100), (1, 36), (9, 100), (10, 100), (11, 75), (1234, 151), (100001, 50)):
        lens = rng.integers(1, 40, size=n)
        names_bytes = int((lens + 1).sum())
        want = sum(1 + int(k) + 1 + rl + 3 + rl + 1 for k in lens)
        assert L.scalce_fastq_text_bytes(rl, n, names_bytes, None) == want
        lib = b"run_7"
        want = sum(len(b"@%s.%d\n" % (lib, i)) + rl + 3 + rl + 1 for i in range(n))
        assert L.scalce_fastq_text_bytes(rl, n, 0, lib) == want


# ---------------------------------------------------------------- parallel gzip reader (scalce_amd/csrc/pargz.hpp)

def _bgzf(data, block=60000):
    """BGZF as bgzip writes it: members of <= 64 KiB with the 'BC' extra field that holds the member's size - 1"""
    import struct
    import zlib
    out = bytearray()
    for a in list(range(0, len(data), block)) + [None]:
        chunk = b"" if a is None else data[a:a + block]     # (the last, empty member is BGZF's end-of-file marker)
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = c.compress(chunk) + c.flush()
        size = 12 + 6 + len(body) + 8
        out += b"\x1f\x8b\x08\x04" + b"\0" * 4 + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, size - 1)
        out += body + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk))
    return bytes(out)


@pytest.mark.parametrize("kind", ["one_member", "many_members", "bgzf", "gz_inside_stored_block", "tiny", "empty_members"])
def test_parallel_gzip_reader(kind, tmp_path):
    """Every kind of gzip file through ParGz on 1, 3 and 16 threads equals Python's gzip: one member (streams through one
    z_stream), many members (threads start at verified member starts), BGZF (the chain of BC fields is walked), a gzip
    file stored verbatim inside another member (a member start that is NOT one: the chain check drops it), members of
    zero bytes.  A cut file is an error, not a short read."""
    import gzip
    import subprocess
    import zlib
    cat = os.path.join(ROOT, "scalce_amd", "bin", "pargz_cat")
    if not os.path.exists(cat):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "scalce_amd", "csrc"), "../bin/pargz_cat"], check=True)
    rng = np.random.default_rng(5)
    text = np.frombuffer(b"ACGTN\n@+IFHD0123", dtype=np.uint8)[rng.integers(0, 16, size=6_000_000)].tobytes()
    if kind == "one_member":
        data = gzip.compress(text, 6)
    elif kind == "many_members":
        data = b"".join(gzip.compress(text[a:a + 300_000], 1 + (a // 300_000) % 9) for a in range(0, len(text), 300_000))
    elif kind == "bgzf":
        data = _bgzf(text)
    elif kind == "gz_inside_stored_block":
        inner = gzip.compress(text[:2_000_000], 6)          # looks like a member wherever it lies
        c = zlib.compressobj(0, zlib.DEFLATED, 31)          # level 0: stored blocks, the inner file verbatim
        data = gzip.compress(text[2_000_000:3_000_000]) + c.compress(inner * 3) + c.flush() + gzip.compress(text[3_000_000:])
        text = text[2_000_000:3_000_000] + inner * 3 + text[3_000_000:]
    elif kind == "tiny":
        text = b"@r\nACGT\n+\nIIII\n"
        data = gzip.compress(text)
    else:
        data = gzip.compress(b"") + gzip.compress(text[:100]) + gzip.compress(b"") * 3 + gzip.compress(text[100:5000]) + gzip.compress(b"")
        text = text[:5000]
    assert gzip.decompress(data) == text
    path = tmp_path / "in.gz"
    open(path, "wb").write(data)
    for threads in (1, 3, 16):
        r = subprocess.run([cat, str(path), str(threads)], capture_output=True)
        assert r.returncode == 0 and r.stdout == text, f"{kind}, {threads} threads: {len(r.stdout)} of {len(text)} bytes, {r.stderr[-200:]}"
    if kind in ("many_members", "bgzf"):
        assert b"windows 1" in r.stderr and b"serial_bytes 0" in r.stderr, r.stderr   # it did go parallel
    if kind != "tiny":
        open(path, "wb").write(data[:-9])
        r = subprocess.run([cat, str(path), "8"], capture_output=True)
        assert r.returncode == 1
    # Behind a complete member anything that does not begin with the gzip magic ends the file, as for zlib's gzread
    # (gz_look ignores trailing garbage: zero padding of block-aligned or tape-written files), whatever XFL the writer set.
    for tail in (b"\0" * 4096, b"\0", b"trailing garbage, not gzip" * 100):
        open(path, "wb").write(data + tail)
        for threads in (1, 8):
            r = subprocess.run([cat, str(path), str(threads)], capture_output=True)
            assert r.returncode == 0 and r.stdout == text, f"{kind} + {len(tail)} tail bytes, {threads} threads: {r.stderr[-200:]}"
    if kind == "many_members":   # a writer that sets another XFL (zlib only writes 0, 2, 4) is still a member where the stream stands
        odd = bytearray(data)
        odd[8] = 3
        open(path, "wb").write(bytes(odd))
        r = subprocess.run([cat, str(path), "8"], capture_output=True)
        assert r.returncode == 0 and r.stdout == text
        # a member too large to be held whole by a window (cap lowered for the test) goes through the one stream, in order
        big = gzip.compress(text[:300_000], 1) * 7 + gzip.compress(text[:2_500_000], 1) + data
        open(path, "wb").write(big)
        env = dict(os.environ, SCALCE_PARGZ_MEMBER_CAP=str(1 << 20))
        r = subprocess.run([cat, str(path), "8"], capture_output=True, env=env)
        assert r.returncode == 0 and r.stdout == gzip.decompress(big) and b"serial_bytes 0" not in r.stderr, r.stderr
        # a member whose body is damaged in the middle of the file is still an error
        bad = bytearray(data)
        for k in range(len(bad) // 2, len(bad) // 2 + 64):
            bad[k] ^= 0x5A
        open(path, "wb").write(bytes(bad))
        r = subprocess.run([cat, str(path), "8"], capture_output=True)
        assert r.returncode == 1

"""CPU: the C restatement (oracle/) against the golden vectors made from the real reference objects."""
import hashlib
import os

import numpy as np
import pytest

import oraclelib as O
from scalce_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["se100", "se100_lossy", "se150_text", "se36_ties", "se100_110k", "se100_f7", "pe150", "pe150_lossy_f2"]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def load_case(name, patterns_blob):
    g = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)
    kw = eval(str(g["kw"]))  # noqa: S307 - literal dict written by make_golden.py
    bases, quals = synth.reads_and_quals(int(g["n"]), int(g["L"]), seed=int(g["seed"]), **kw)
    trie = O.Trie(text=str(g["ptxt"]).encode()) if "ptxt" in g else O.Trie(blob=patterns_blob)
    return g, bases, quals, trie


def name_stream(n, prefix, suffix):
    """[u8 len][chars] per read, input order: what output_name (names.cpp:48-62) makes of the generator's names."""
    out = bytearray()
    for i in range(n):
        nm = (prefix + str(i) + suffix).encode()
        out.append(len(nm))
        out += nm
    return np.frombuffer(bytes(out), dtype=np.uint8)


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_vectors(name, patterns_blob):
    g, bases, quals, trie = load_case(name, patterns_blob)
    pat, end = trie.tokenize(bases)
    tok = np.stack([pat, end], axis=1).astype(np.int32)
    assert sha(tok) == str(g["sha_tok"])
    perm = trie.order(bases, pat, end)
    assert sha(perm.astype(np.int64)) == str(g["sha_order"])
    lens = trie.pattern_lens()
    packed = []
    for r in range(len(bases)):
        lvl = int(lens[pat[r]]) if pat[r] >= 0 else 0
        e = int(end[r])
        packed.append(O.pack_read(bases[r], e - lvl if e else 0, lvl if e else 0))
    assert sha(np.concatenate(packed)) == str(g["sha_packed"])
    if "lut" in g:
        off, vals = int(g["lut"][0]), g["lut"][1:]
    else:
        off, vals = 33, np.arange(128)
    qp, f4 = O.quality_stream(quals, bases, off, vals)
    assert sha(qp) == str(g["sha_qual"])
    assert sha(f4) == str(g["sha_freq4"])
    factor = int(g["factor"])
    table = O.ac_scale(f4, factor)  # compress.cpp:297-313
    assert sha(table) == str(g["sha_table"])
    st = O.AcStat(table)
    enc = st.encode_stream(qp[perm].reshape(-1))
    assert len(enc) == int(g["ac_len"])
    assert sha(enc) == str(g["sha_ac"])
    assert sha(name_stream(len(bases), "p." if "seed2" in g else "s.", "/1" if "seed2" in g else "")) == str(g["sha_names"])
    if "seed2" in g:  # mate 2: whole read packed, own quality model and counters (qualities.cpp:179 prev[1])
        bases2, quals2 = synth.reads_and_quals(int(g["n"]), int(g["L"]), seed=int(g["seed2"]), **eval(str(g["kw"])))  # noqa: S307
        assert sha(np.concatenate([O.pack_read(r, 0, 0) for r in bases2])) == str(g["sha_packed2"])
        off2, vals2 = (int(g["lut2"][0]), g["lut2"][1:]) if "lut2" in g else (33, np.arange(128))
        qp2, f42 = O.quality_stream(quals2, bases2, off2, vals2)
        assert sha(qp2) == str(g["sha_qual2"])
        assert sha(f42) == str(g["sha_freq4_2"])
        table2 = O.ac_scale(f42, factor)
        assert sha(table2) == str(g["sha_table2"])
        enc2 = O.AcStat(table2).encode_stream(qp2[perm].reshape(-1))
        assert len(enc2) == int(g["ac2_len"]) and sha(enc2) == str(g["sha_ac2"])
        assert (enc2[:4096] == g["ac2_head"]).all()
    if "tok" in g:  # element-wise views for debuggability
        assert (tok == g["tok"]).all()
        assert (perm == g["order"]).all()
        assert (trie.pattern_ids() == g["ids"]).all()
        assert (enc[:4096] == g["ac_head"]).all()
        nz = np.flatnonzero(f4 != 1)
        assert (nz == g["freq4_idx"]).all() and (f4[nz].astype(np.int64) == g["freq4_val"]).all()


def test_ac_roundtrip_blocks(patterns_blob):
    g, bases, quals, trie = load_case("se100", patterns_blob)
    qp, f4 = O.quality_stream(quals, bases, 33, np.arange(128))
    st = O.AcStat(O.ac_scale(f4, 1))
    for n in (2, 3, 17, 1000, 100000):
        sym = qp.reshape(-1)[:n]
        enc = st.encode_block(sym)
        assert (st.decode_block(enc, n) == sym).all()


def test_lossy_lut_known_answer():
    """SURVEY.md 8(a-10) [verified on the reference]: -p 30 on the N(30,8) generator gives
    {2..5}->0, {28..33}->30, {38..40}->40, everything else identity; offset 33."""
    _, q = synth.reads_and_quals(100000, 100)
    off, vals = O.qmap_init(np.bincount(q.reshape(-1), minlength=128), 30)
    assert off == 33
    want = {c: c for c in range(128)}
    for c in range(0, 6):
        want[33 + c] = 33  # also 0,1: >30 % error, never seen in the data
    for c in range(28, 34):
        want[33 + c] = 33 + 30
    for c in range(38, 41):
        want[33 + c] = 33 + 40
    seen = np.flatnonzero(np.bincount(q.reshape(-1), minlength=128))
    for c in seen:
        assert vals[c] == want[c], (c - 33, vals[c] - 33)
    off0, vals0 = O.qmap_init(np.bincount(q.reshape(-1), minlength=128), 0)
    assert off0 == 33 and (vals0 == np.arange(128)).all()
    off64, _ = O.qmap_init(np.bincount(q.reshape(-1) + 31, minlength=128), 0)
    assert off64 == 64


def test_file_roundtrip_and_known_sizes(tmp_path, patterns_blob):
    """Whole-file restatement: compress + decompress is the identity on canonical FASTQ, and the
    stream sizes on the first 100 000 reads of the BASELINE.md generator are stable."""
    pbin = os.path.join(GOLD, "patterns.bin")
    for paired, extra in ((False, []), (False, ["-p", "30"]), (False, ["-A"]), (False, ["-n", "lib"]),
                          (True, ["-r"]), (False, ["-B", "400000"]), (False, ["-c", "gz"])):
        d = tmp_path / ("c" + "_".join(extra).replace("-", ""))
        d.mkdir()
        b1, q1 = synth.write_fastq(str(d / "in_1.fq"), 5000, 100, seed=7, n_frac=0.005, dup_frac=0.1,
                                   paired_suffix="/1" if paired else None)
        if paired:
            synth.write_fastq(str(d / "in_2.fq"), 5000, 100, seed=8, paired_suffix="/2")
        O.orc_cli("compress", pbin, d / "in_1.fq", d / "out", *extra)
        dextra = [x for x in extra if x in ("-r",)] + (["-n", "lib"] if "-n" in extra else [])
        O.orc_cli("decompress", pbin, d / "out_1.scalcen", d / "back", *dextra)
        for m in ((1, 2) if paired else (1,)):
            src = open(d / f"in_{m}.fq", "rb").read().split(b"\n")
            got = open(d / f"back_{m}.fastq", "rb").read().split(b"\n")
            def canon(L):  # an N base is stored with quality 0 and comes back as '!' (qualities.cpp:183)
                out = []
                for i in range(0, len(L) - 1, 4):
                    q = bytes(33 if b == 78 else c for b, c in zip(L[i + 1], L[i + 3]))
                    out.append((L[i], L[i + 1], L[i + 2], q))
                return out
            key = lambda L: sorted(canon(L))
            if "-n" in extra:  # names are regenerated as lib.<k>
                strip = lambda L: sorted(x[1:] for x in canon(L))
                assert strip(src) == strip(got)
            elif "-p" in extra:
                assert len(src) == len(got)
            else:
                # a base whose quality maps to 0 decodes as N; the generator's floor is q=2
                assert key(src) == key(got)

"""Device-side synthetic FASTQ (bench.py and the full-size property tests).

Same record shape and distributions as scalce_amd.synth: ``@s.<i>`` names, uniform ACGT,
qualities clip(round(N(30, 8)), 2, 40) + 33, bare ``+``.  The random stream is torch's, so the bytes differ
from the numpy generator; the shard is built in chunks directly in HBM.
"""
import torch


def fastq_on_device(n, length, device, seed=20261003, first_index=0, chunk=1 << 21, return_parts=False):
    """Return a uint8 tensor holding n FASTQ records (16-byte aligned storage, exact length)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    L = length
    idx_all_digits = len(str(first_index + max(n - 1, 0)))
    # total size: per record 2L + 8 + digits(i)
    total = 0
    bounds = []
    a = 0
    while a < n:
        b = min(n, a + chunk)
        i = torch.arange(first_index + a, first_index + b, device=device, dtype=torch.int64)
        d = torch.ones_like(i)
        p = 10
        for _ in range(idx_all_digits):
            d += (i >= p).to(torch.int64)
            p *= 10
        rec = 2 * L + 8 + d
        bounds.append((a, b, total, int(rec.sum().item())))
        total += bounds[-1][3]
        a = b
    text = torch.empty(total + 64, dtype=torch.uint8, device=device)
    acgt = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=device)
    ar = torch.arange(L, device=device, dtype=torch.int64)
    parts = []
    for a, b, base, _sz in bounds:
        m = b - a
        i = torch.arange(first_index + a, first_index + b, device=device, dtype=torch.int64)
        d = torch.ones_like(i)
        p = 10
        for _ in range(idx_all_digits):
            d += (i >= p).to(torch.int64)
            p *= 10
        rec = 2 * L + 8 + d
        off = base + torch.cumsum(rec, 0) - rec
        text[off] = 64  # '@'
        text[off + 1] = 115  # 's'
        text[off + 2] = 46  # '.'
        p10 = 1
        for k in range(idx_all_digits):  # k-th digit from the right
            sel = d > k
            pos = off[sel] + 2 + d[sel] - k
            text[pos] = ((i[sel] // p10) % 10 + 48).to(torch.uint8)
            p10 *= 10
        s0 = off + 3 + d
        text[s0] = 10
        bases = acgt[torch.randint(0, 4, (m, L), device=device, generator=g)]
        q = torch.clamp(torch.round(torch.randn((m, L), device=device, generator=g) * 8.0 + 30.0), 2, 40).to(torch.uint8) + 33
        pos = (s0 + 1).unsqueeze(1) + ar.unsqueeze(0)
        text[pos] = bases
        e = s0 + 1 + L
        text[e] = 10
        text[e + 1] = 43
        text[e + 2] = 10
        text[pos + (L + 3)] = q
        text[e + 3 + L] = 10
        if return_parts:
            parts.append((bases.cpu(), q.cpu()))
        del pos, bases, q
    out = text[:total]
    return (out, parts) if return_parts else out

"""Integer model of the bf16 exact route of K1 (quantization_analysis_amd/csrc/mtq_fast.hip), one shared-exponent group
at a time, against the oracle's literal stats definition: the exactness argument of the kernel (every main-class value is
an integer multiple of 2^(E-149); group sums are small integers times one power of two) checked in Python integers on the
CPU.  Groups with elements more than 15 binades below the maximum or with E outside [80,180] are the kernel's tail /
fix-up cases and are skipped here (the GPU tests cover them)."""
import numpy as np

from oracle import mtq_oracle as orc
from tests.inputs import gen


def group_terms(h):
    """h: uint16[16] bf16 bits. returns dict of exact python-int/fraction based float64 terms or None if not fast."""
    a_ = h & 0x7FFF
    e = a_ >> 7
    E = int(e.max())
    d = E - e
    if not (80 <= E <= 180) or d.max() > 15:
        return None
    m = (a_ & 0x7F) | 0x80
    sgn = np.where(h >> 15, -1, 1).astype(np.int64)
    a = np.where(d <= 7, m.astype(np.int64) << np.maximum(7 - d, 0), 0)
    b = np.where(d >= 8, m.astype(np.int64) << np.maximum(15 - d, 0), 0)
    S1 = 2.0 ** (E - 149)
    S2 = 2.0 ** (2 * E - 298)
    out = {}
    out["sx"] = float(256 * int((sgn * a).sum()) + int((sgn * b).sum())) * S1
    out["sx2"] = float(65536 * int((a * a).sum()) + int((b * b).sum())) * S2
    sb = int(b.sum()); bmax = int(b.max())
    for mb, name in ((7, "bfp8"), (3, "bfp4"), (1, "bfp2")):
        sh = 15 - mb
        G = 1 << sh
        t = a + (G // 2 - 1) + ((a >> sh) & 1)
        y = np.minimum(t & ~(G - 1), ((1 << mb) - 1) * G)
        q = y >> sh
        dl = np.abs(a - y)
        out[name] = (
            float(int((sgn * y).sum())) * 2.0 ** (E - 141),
            float(int((q * q).sum())) * 2.0 ** (2 * (E - 126 - mb)),
            float(int((a * q).sum())) * 2.0 ** (2 * E - 267 - mb),
            float(256 * int(dl.sum()) + sb) * S1,
            float(max(256 * int(dl.max()), bmax)) * S1,
        )
    out["bf16"] = (out["sx"], out["sx2"], out["sx2"], 0.0, 0.0)
    return out


def test_integer_model_equals_literal_sums():
    ALL = ["bf16", "bfp8", "bfp4", "bfp2"]
    nfast = ntot = 0
    for kind, seed in (("normal_bf16", 1), ("heavy_bf16", 2), ("heavy_bf16", 3)):
        x = gen(kind, seed, (64, 256))
        if seed == 3:
            x = x * np.float32(2.0 ** 20)
        hb = (x.view(np.uint32) >> 16).astype(np.uint16)
        for r in range(x.shape[0]):
            for c0 in range(0, x.shape[1], 16):
                ntot += 1
                g = group_terms(hb[r, c0:c0 + 16])
                if g is None:
                    continue
                nfast += 1
                # oracle stats for a "tile" holding only this group: put the group in a 1x16 matrix
                st = orc.tile_stats(x[r:r + 1, c0:c0 + 16], ALL)[0]
                want = {"sx": st[0], "sx2": st[1]}
                assert g["sx"] == st[0] and g["sx2"] == st[1], (kind, r, c0, g["sx"], st[0], g["sx2"], st[1])
                for k, f in enumerate(ALL):
                    blk = tuple(st[2 + 5 * k: 7 + 5 * k])
                    assert g[f] == blk, (kind, r, c0, f, g[f], blk)
    assert ntot == 3072 and nfast >= 3000  # nearly every group of these inputs is on the exact route

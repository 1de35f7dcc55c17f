"""Integer model of the bf16 fast path of K1 (one shared-exponent group) checked against the oracle's
stats definition.  Development aid for quantization_analysis_amd/csrc/mtq_fast.hip (not shipped code)."""
import sys
import numpy as np
sys.path.insert(0, "/root/repo")
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


def main():
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
    print("fast groups", nfast, "of", ntot, "all exact")


main()

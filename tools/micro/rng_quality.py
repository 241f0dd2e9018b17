"""Statistical screen of the dropout mixer (csrc/common.h rng_u24_from_x0) against the 'lowbias32' finaliser it replaced in round 5:
keep rate at p = 0.1, autocorrelation of the masks at lags 1, 2 and N = 1024 (neighbours along a row and along a column of an
activation matrix), chi^2 of the 24-bit values over 256 bins, agreement of the masks of indices 2^24 apart (0.82 = independent: the
mixer must not have a period of 2^24 although its multiplier sees 24 bits), correlation between the masks of consecutive seeds.
CPU only (numpy emulation of the two mixers):   python tools/micro/rng_quality.py"""
import numpy as np
M32 = np.uint64(0xFFFFFFFF)


def lowbias(x, sh):
    x = x ^ sh
    x ^= x >> np.uint32(16); x = (x.astype(np.uint64) * np.uint64(0x7FEB352D) & M32).astype(np.uint32)
    x ^= x >> np.uint32(15); x = (x.astype(np.uint64) * np.uint64(0x846CA68B) & M32).astype(np.uint32)
    x ^= x >> np.uint32(16)
    return x >> np.uint32(8)


def mul24(a, c):
    return ((a & np.uint32(0xFFFFFF)).astype(np.uint64) * np.uint64(c) & M32).astype(np.uint32)


def mix24(x, sh):
    x = x ^ sh
    x ^= x >> np.uint32(16)
    y = mul24(x, 0x6B43A9)
    y ^= y >> np.uint32(15)
    y = (mul24(y, 0x52DCE7).astype(np.uint64) + x.astype(np.uint64) & M32).astype(np.uint32)
    y ^= y >> np.uint32(14)
    return y >> np.uint32(8)


def u24(fn, idx, seed):
    x0 = ((idx.astype(np.uint64) * np.uint64(0x9E3779B1) + np.uint64(seed & 0xFFFFFFFF)) & M32).astype(np.uint32)
    return fn(x0, np.uint32(seed >> 32))


if __name__ == '__main__':
    rng = np.random.default_rng(0)
    for name, fn in (('lowbias32 (rounds 1-4)', lowbias), ('mix24 (round 5)', mix24)):
        print(name)
        for seed in (0x5EED00010000, 0x123456789ABC0000, 0x5EED00020000 + (7 << 32)):
            N = 1024
            idx = np.arange(0, 1 << 22, dtype=np.uint64) + np.uint64(rng.integers(0, 1 << 30))
            u = u24(fn, idx, seed).astype(np.float64) / 2 ** 24
            keep = (u >= 0.1).astype(np.float64)
            h = np.bincount((u * 256).astype(int), minlength=256)
            chi = ((h - len(u) / 256) ** 2 / (len(u) / 256)).sum()
            u2 = u24(fn, idx + np.uint64(1 << 24), seed).astype(np.float64) / 2 ** 24
            print(f'  seed {seed:x}: keep {keep.mean():.5f}  corr lag 1 {np.corrcoef(keep[:-1], keep[1:])[0, 1]:+.4f} lag 2 '
                  f'{np.corrcoef(keep[:-2], keep[2:])[0, 1]:+.4f} lag N {np.corrcoef(keep[:-N], keep[N:])[0, 1]:+.4f}  chi2(255) {chi:.0f}  '
                  f'agree with idx + 2^24: {np.mean((u2 >= 0.1) == (u >= 0.1)):.4f}')
        idx = np.arange(0, 1 << 20, dtype=np.uint64)
        a = (u24(fn, idx, 0x5EED00010000) >= 0.1 * 2 ** 24).astype(float)
        b = (u24(fn, idx, 0x5EED00020000) >= 0.1 * 2 ** 24).astype(float)
        print(f'  correlation between the masks of consecutive seeds: {np.corrcoef(a, b)[0, 1]:+.4f}')

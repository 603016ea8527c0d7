"""Quality of the 24-bit-multiply hash (nvt_common.hpp: mul24_hash) against fmix32, on the CPU:
keys of the first ids of a scrambled / dense / strided / random column
  * hot image: how many find a free slot of the 4096 x 2 image (first come),
  * LDS-resident counting: longest probe chain of a 16384-slot linear-probing table.
python tools/hash_quality.py"""
import numpy as np


def fmix32(k):
    h = k.astype(np.uint64) & 0xFFFFFFFF
    h ^= h >> 16
    h = (h * 0x85EBCA6B) & 0xFFFFFFFF
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & 0xFFFFFFFF
    h ^= h >> 16
    return h


def mul24(k):
    k = k.astype(np.uint64) & 0xFFFFFFFF
    a = (((k ^ (k >> 7)) & 0xFFFFFF) * 0x9E3779) & 0xFFFFFFFF
    b = (((k >> 8) & 0xFFFFFF) * 0x5BD1E9) & 0xFFFFFFFF
    return (a + b) & 0xFFFFFFFF


def placed(b, width=2, nb=4096):
    return int(np.minimum(np.bincount(b.astype(np.int64), minlength=nb), width).sum())


def longest_chain(home, slots=16384):
    """Linear probing, keys inserted in order: the largest displacement."""
    table = np.zeros(slots, dtype=bool)
    worst = 0
    for h in home.astype(np.int64):
        d = 0
        while table[(h + d) % slots]:
            d += 1
        table[(h + d) % slots] = True
        worst = max(worst, d)
    return worst


def key_sets(n):
    x = np.arange(1, n + 1, dtype=np.uint64)
    return {
        "scrambled": (x * 2654435761 + 97 * 3) % (2**31),
        "dense": x,
        "stride 64": x * 64,
        "stride 4096": (x * 4096) % (2**32),
        "negative": (2**32 - x * 7919) % (2**32),
        "random": np.random.default_rng(0).integers(0, 2**32, n).astype(np.uint64),
    }


if __name__ == "__main__":
    for name, k in key_sets(8192).items():
        print("hot image, 8192 keys  %-12s fmix32 %5d   mul24 %5d" % (
            name, placed((fmix32(k) >> 13) & 4095), placed((mul24(k) >> 20) & 4095)))
    for n in (4096, 11000):
        for name, k in key_sets(n).items():
            print("LDS table, %5d keys  %-12s fmix32 chain %4d   mul24 chain %4d" % (
                n, name, longest_chain((fmix32(k) >> 17) & 16383), longest_chain((mul24(k) >> 18) & 16383)))

"""Seed derivation for the counter-based dropout generator (csrc/vlfb_head.hip): one 64-bit seed
per (experiment seed, dropout blob, iteration), so every replica/iteration/op draws an
independent, reproducible mask."""

_MASK = (1 << 64) - 1


def _fnv1a64(text):
    h = 0xCBF29CE484222325
    for ch in text.encode("utf-8"):
        h = ((h ^ ch) * 0x100000001B3) & _MASK
    return h


def dropout_seed(base_seed, blob_name, iteration):
    return (_fnv1a64(blob_name) ^ ((int(base_seed) * 0x9E3779B97F4A7C15 + int(iteration)) & _MASK)) & _MASK

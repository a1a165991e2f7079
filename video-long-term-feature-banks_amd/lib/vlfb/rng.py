"""Seed derivation for the counter-based dropout generator (csrc/vlfb_head.hip): one 64-bit seed
per (experiment seed, dropout blob, iteration, data-parallel replica), so every replica /
iteration / op draws an independent, reproducible mask."""

_MASK = (1 << 64) - 1


def _fnv1a64(text):
    h = 0xCBF29CE484222325
    for ch in text.encode("utf-8"):
        h = ((h ^ ch) * 0x100000001B3) & _MASK
    return h


def dropout_seed(base_seed, blob_name, iteration, replica=0):
    """`replica` = data-parallel rank: the reference's per-GPU Dropout ops draw independent masks
    (one Caffe2 RNG per device), so replicas must not share a mask; replica 0 keeps the one-GPU stream"""
    mixed = (int(base_seed) * 0x9E3779B97F4A7C15 + int(iteration) + int(replica) * 0xD1B54A32D192ED03) & _MASK
    return (_fnv1a64(blob_name) ^ mixed) & _MASK

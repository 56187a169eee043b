"""Deterministic synthetic RGB frames (host twin of csrc/khg_synth.h, integer arithmetic only)."""
import numpy as np


def _tri(t, p):
    m = t % np.uint32(2 * p)
    return np.where(m < p, m, np.uint32(2 * p) - m).astype(np.uint32)


def _hash32(v):
    v = v.astype(np.uint32)
    v ^= v >> np.uint32(15)
    v *= np.uint32(0x2C1B3C6D)
    v ^= v >> np.uint32(12)
    v *= np.uint32(0x297A2D39)
    v ^= v >> np.uint32(15)
    return v


def synth_frame(rows, cols, seed, index):
    """(3, rows, cols) uint8 planar RGB, identical to khg_synth_dev on the device."""
    with np.errstate(over="ignore"):
        y = np.arange(rows, dtype=np.uint32)[:, None]
        x = np.arange(cols, dtype=np.uint32)[None, :]
        idx = np.uint32(index)
        key = np.uint32((int(seed) * 1000003 + int(index)) & 0xFFFFFFFF)
        luma = (np.uint32(36) + _tri(x + np.uint32(3) * idx, 97) * np.uint32(64) // np.uint32(97)
                + _tri(y + np.uint32(5) * idx, 131) * np.uint32(64) // np.uint32(131)
                + _tri(x + y, 251) * np.uint32(64) // np.uint32(251))
        h = _hash32((y * np.uint32(cols) + x) * np.uint32(0x9E3779B1) ^ (key * np.uint32(0x85EBCA6B)))
        n = (h % np.uint32(13)).astype(np.int32) - 6
        l = luma.astype(np.int32) + n
        r = np.clip(l + 12, 0, 255)
        g = np.clip(l, 0, 255)
        b = np.clip(l - 9 + (_tri(y, 173) * np.uint32(24) // np.uint32(173)).astype(np.int32), 0, 255)
        return np.ascontiguousarray(np.stack([r, g, b]).astype(np.uint8))

// khg_synth.h -- deterministic synthetic RGB frames for benchmarks and tests.
//
// Integer arithmetic only, so the device generator and the host generator
// (khugle_amd/synth.py) agree bit for bit.  A frame is a smooth luma field (three
// triangle waves that drift with the frame index) plus +-6 of hashed per-pixel noise,
// with fixed per-channel offsets so the channels stay correlated (independent RGB noise
// would be destroyed by the decimating 4:2:0 of Colorspace.h:15-22).
#pragma once
#include "khg_common.h"

KHG_DEV uint32_t khg_tri(uint32_t t, uint32_t p)
{
    uint32_t m = t % (2 * p);
    return m < p ? m : 2 * p - m;  // 0..p
}

KHG_DEV uint32_t khg_hash32(uint32_t v)
{
    v ^= v >> 15;
    v *= 0x2C1B3C6Du;
    v ^= v >> 12;
    v *= 0x297A2D39u;
    v ^= v >> 15;
    return v;
}

KHG_DEV int khg_clamp255(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }

KHG_DEV void khg_synth_pixel(uint32_t y, uint32_t x, uint32_t cols, uint32_t seed, uint32_t index,
                             uint8_t* r, uint8_t* g, uint8_t* b)
{
    uint32_t key = seed * 1000003u + index;
    uint32_t luma = 36 + khg_tri(x + 3 * index, 97) * 64 / 97 + khg_tri(y + 5 * index, 131) * 64 / 131 +
                    khg_tri(x + y, 251) * 64 / 251;
    uint32_t h = khg_hash32((y * cols + x) * 0x9E3779B1u ^ key * 0x85EBCA6Bu);
    int n = (int)(h % 13u) - 6;
    int l = (int)luma + n;
    *r = (uint8_t)khg_clamp255(l + 12);
    *g = (uint8_t)khg_clamp255(l);
    *b = (uint8_t)khg_clamp255(l - 9 + (int)(khg_tri(y, 173) * 24 / 173));
}

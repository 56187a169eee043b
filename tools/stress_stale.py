"""Developer probe: the mixed-size batch test's scenario, repeated with fresh Batch objects, with a report of WHERE
a wrong decode differs and whether it equals a decode that used the previous frame's level-1 LL quadrant."""
import os
import sys
import time

import numpy as np

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for p in ("image-compression_amd/python", "oracle", "tests"):
    sys.path.insert(0, os.path.join(R, p))
import oracle_py as O
from khugle_amd.batch import Batch
from khugle_amd.synth import synth_frame

H, W = 264, 520
L = O.lib()


def stale_ll_decode(q0, stale_from_rgb):
    """Decode of symbols q0 with the luma level-1 LL quadrant replaced by the forward LL of another frame."""
    y, cb, cr = O.rgb_to_ycbcr(stale_from_rgb)
    c = np.ascontiguousarray(y.copy())
    L.khgo_fwt2d(c, W, H, W)
    ll = c[:H // 2, :W // 2].copy()
    planes = []
    for p, q in enumerate(q0):
        h, w = q.shape
        div = np.zeros((h, w))
        L.khgo_divisor_plane(H, W, 1, 1 if p else 0, div)
        c = np.ascontiguousarray(q.astype(np.float64) * div)
        L.khgo_iwt2d(c, w, h // 2, w // 2)
        if p == 0:
            c[:h // 2, :w // 2] = ll
        L.khgo_iwt2d(c, w, h, w)
        planes.append(c)
    out = np.zeros((3, H, W), np.uint8)
    L.khgo_ycbcr_to_rgb(planes[0], planes[1], planes[2], H, W, out[0], out[1], out[2])
    return out


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    frames = [synth_frame(H, W, 2025, i) for i in range(7)]
    comps = [O.encode(f, 1) for f in frames]
    want = [O.decode(c) for c in comps]
    order = [0, 1, 2, 3, 4, 5, 6]
    bad = 0
    t0 = time.time()
    for it in range(iters):
        b = Batch(512, 520, depth=3)
        got_c = b.encode(frames, 1)
        if got_c != comps:
            print("iter %d: ENCODE bytes differ" % it, flush=True)
            bad += 1
        outs = b.decode([comps[i] for i in order])
        for k, (g, i) in enumerate(zip(outs, order)):
            if not np.array_equal(g, want[i]):
                bad += 1
                d = np.argwhere(g != want[i])
                print("iter %d file %d (slot %d): %d differing values, channels %s rows %d..%d cols %d..%d" % (
                    it, k, k % 3, len(d), sorted(set(d[:, 0])), d[:, 1].min(), d[:, 1].max(), d[:, 2].min(), d[:, 2].max()),
                    flush=True)
                q0 = O.symbols(frames[i], 1)
                for j in range(7):
                    h1 = stale_ll_decode(q0, frames[j])
                    same = (g == h1)
                    print("   vs stale-LL-of-frame-%d: %.1f %% of pixels equal; rows fully equal: %s" % (
                        j, 100.0 * same.mean(), np.flatnonzero(same.all(axis=(0, 2)))[:40]), flush=True)
                rows_bad = np.flatnonzero((g != want[i]).any(axis=(0, 2)))
                cols_bad = np.flatnonzero((g != want[i]).any(axis=(0, 1)))
                print("   bad rows %s\n   bad cols %s" % (rows_bad, cols_bad), flush=True)
        b.close()
        if time.time() - t0 > 90:
            print("time limit after %d iterations" % (it + 1))
            break
    print("bad", bad, "iters", it + 1)


main()

"""Developer probe: repeat the mixed-size batch decode and report where results differ from the oracle."""
import os, sys
import numpy as np
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(R, "image-compression_amd", "python")); sys.path.insert(0, os.path.join(R, "oracle")); sys.path.insert(0, os.path.join(R, "tests"))
import oracle_py as O
from khugle_amd.batch import Batch
from khugle_amd.synth import synth_frame
from khugle_amd import read_bmp
G = os.path.join(R, "tests", "golden", "run")
cat = read_bmp(os.path.join(G, "cat.bmp")); lena = read_bmp(os.path.join(G, "lena.bmp")) if os.path.exists(os.path.join(G, "lena.bmp")) else cat
b = Batch(512, 520, depth=3)
frames = [synth_frame(264, 520, 2025, i) for i in range(7)]
comps = [O.encode(f, 1) for f in frames]
mixed = comps[:3] + [O.encode(cat, 0), O.encode(lena, 2)] + comps[3:]
want = [O.decode(c) for c in mixed]
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    outs = b.decode(mixed)
    for i, (g, w) in enumerate(zip(outs, want)):
        if not np.array_equal(g, w):
            d = np.argwhere(g != w)
            bad += 1
            print("iter %d image %d shape %s: %d differing values; channels %s rows %d..%d cols %d..%d" % (
                it, i, g.shape, len(d), sorted(set(d[:, 0])), d[:, 1].min(), d[:, 1].max(), d[:, 2].min(), d[:, 2].max()), flush=True)
print("bad", bad)

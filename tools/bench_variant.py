"""Developer aid: bench.py against an alternative build of the library (KHG_LIB=path), short form."""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "image-compression_amd", "python"))
sys.path.insert(0, ROOT)
import khugle_amd.lib
if os.environ.get("KHG_LIB"):
    khugle_amd.lib.LIB_PATH = os.environ["KHG_LIB"]
import json
import io
import contextlib
import bench

sys.argv = ["bench.py", "--no-cpu-baseline", "--no-square", "--no-host-pipeline", "--no-isolated-kernel", "--no-format2"] + sys.argv[1:]
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
line = json.loads(buf.getvalue().strip().splitlines()[-1])
print(os.path.basename(khugle_amd.lib.LIB_PATH), "value", line["value"], "encode", line["encode_mp_s"], "decode",
      line["decode_mp_s"], "parity", line["parity_spot_check"])

"""bench.py against a VARIANT build of the library (same ABI), for same-box A/B runs: LADIFF_LIB=ladiff_amd/libladiff_hip_x.so python
scripts/bench_with_lib.py --config headline ...   The product bench never reads LADIFF_LIB; this wrapper sets the path and calls it."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ladiff_amd import _lib
if os.environ.get("LADIFF_LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, os.environ["LADIFF_LIB"])
import bench
bench.main()

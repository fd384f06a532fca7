"""bench.py against the lab library (environment knobs honoured): python tools/bench_lab.py [bench.py args]
MMS_LAB_LIB=<path> picks another lab build (e.g. csrc/libmmscore_lab_n5.so)."""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kddcup_2020_multimodalitiesrecall_2nd_place_amd import lib
lib.load(os.environ.get("MMS_LAB_LIB") or lib.LAB_LIB_PATH)
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")

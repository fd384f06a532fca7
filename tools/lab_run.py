"""Run any tool / script against the lab library (environment knobs honoured): python tools/lab_run.py tools/small_batch.py zk 256"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kddcup_2020_multimodalitiesrecall_2nd_place_amd import lib
lib.load(os.environ.get("MMS_LAB_LIB") or lib.LAB_LIB_PATH)
sys.argv = sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")

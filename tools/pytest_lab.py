"""Run (a selection of) the GPU tests against a lab build: MMS_LAB_LIB=<path> python tools/pytest_lab.py [pytest args]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kddcup_2020_multimodalitiesrecall_2nd_place_amd import lib
lib.load(os.environ.get("MMS_LAB_LIB") or lib.LAB_LIB_PATH)
import pytest
sys.exit(pytest.main(sys.argv[1:]))

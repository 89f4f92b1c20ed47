"""Run bench.py against an alternative build of the library (tuning experiments): BB_LIB=<path to .so>."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from blackbird_amd import _lib
_lib.LIB_PATH = os.path.abspath(os.environ["BB_LIB"])
import bench
bench.main()

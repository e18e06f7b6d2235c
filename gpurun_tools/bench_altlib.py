"""Diagnostic: run bench.py against another build of the library (A/B of a compile-time variant on one box):
python gpurun_tools/bench_altlib.py gpurun_tools/libg2v_alt.so [bench.py arguments]"""
import os, sys, runpy
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from gesture2vec_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(root, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")

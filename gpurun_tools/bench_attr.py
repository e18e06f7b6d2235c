"""bench.py with engine attributes overridden from the environment (A/B tooling only; the product reads no such variable):
G2V_SIDE_EARLY / G2V_MERGED_PREPARE = 0 | 1 -> VQVAEEngine.side_early / .merged_prepare."""
import os, runpy, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gesture2vec_amd import engine as E
_init = E.VQVAEEngine.__init__
def init(self, *a, **k):
    _init(self, *a, **k)
    for attr, var in (("side_early", "G2V_SIDE_EARLY"), ("merged_prepare", "G2V_MERGED_PREPARE")):
        if var in os.environ:
            setattr(self, attr, os.environ[var] != "0")
E.VQVAEEngine.__init__ = init
sys.argv[0] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
runpy.run_path(sys.argv[0], run_name="__main__")

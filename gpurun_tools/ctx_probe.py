import sys; sys.path.insert(0, '.')
import torch
from gesture2vec_amd import _lib
lib = _lib.load()
c = _lib.Context()
print("cluster_ok default", lib.g2v_dec_rollout_cluster_ok(128, 40, 200), "tiles", lib.g2v_dec_rollout_tiles_per_workgroup(256, 135, 64))
c.set(1, 0); c.set(2, 0)
with c:
    print("bound persist", lib.g2v_ctx_get_option(None, 1), "cluster_ok", lib.g2v_dec_rollout_cluster_ok(128, 40, 200), "tiles", lib.g2v_dec_rollout_tiles_per_workgroup(256, 135, 64))
print("after", lib.g2v_dec_rollout_cluster_ok(128, 40, 200))
lib.g2v_dec_rollout_set_persistent(0)
print("default off", lib.g2v_dec_rollout_cluster_ok(128, 40, 200)); lib.g2v_dec_rollout_set_persistent(1)

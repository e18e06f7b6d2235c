"""Worker of tests/test_gpu_dp_nccl.py: one rank of a 2-GPU data-parallel run over RCCL ("nccl" backend), started by
torch.distributed.run.  Runs the PRODUCT path (VQVAEEngine.train_step with GradStatsAllReduce, explicit keep-masks so that the
CPU oracle can follow) for N steps on this rank's shard and writes its final state + the reduced comm buffer's checksum."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    out_dir, B, T, D, H, K, n_steps = sys.argv[1], *[int(a) for a in sys.argv[2:8]]
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)       # RCCL on ROCm
    from oracle import g2v_oracle as O          # initial state + shard inputs only (the checker compares afterwards)
    from test_gpu_dp_engine import _shard
    from gesture2vec_amd.engine import VQVAEEngine
    from gesture2vec_amd.dp import GradStatsAllReduce, broadcast_state
    sd = O.init_vqvae_state(D, H, 2, K, seed=11 + 100 * rank)      # ranks start DIFFERENT on purpose ...
    eng = VQVAEEngine(D, H, 2, K, T, beta=0.25, dropout_prob=0.0, device=dev)
    for name, _ in eng.layout:
        eng.view(name).copy_(sd[name])
    eng.vq_pre_w.copy_(sd["vq_layer.pre_linear.weight"]); eng.vq_pre_b.copy_(sd["vq_layer.pre_linear.bias"])
    eng.codebook.copy_(sd["vq_layer._embedding.weight"]); eng.ema_w.copy_(sd["vq_layer._ema_w"])
    eng.ema_cs.copy_(sd["vq_layer._ema_cluster_size"])
    broadcast_state([eng.flat, eng.codebook, eng.ema_w, eng.ema_cs, eng.vq_pre_w, eng.vq_pre_b, eng.bn_rm, eng.bn_rv])   # ... rank 0's
    reduce_fn = GradStatsAllReduce()
    sums = []
    for step in range(n_steps):
        x, masks = _shard(rank, step, B, T, D, H, 0.0)
        eng.set_masks(B, masks["dec"].to(dev))
        xd = x.to(dev)
        eng.train_step(xd, xd, lr=5e-4, w_l1=5.0, w_cont=0.1, w_var=0.5, draw_masks=False, reduce_fn=reduce_fn, world=world)
        sums.append(float(eng.comm.double().sum()))               # the REDUCED buffer: must be identical on every rank
    torch.cuda.synchronize()
    assert eng.lib.g2v_dec_rollout_persist_fault(0) == 0
    torch.save({"flat": eng.flat.cpu(), "codebook": eng.codebook.cpu(), "ema_w": eng.ema_w.cpu(), "ema_cs": eng.ema_cs.cpu(),
                "comm_sums": sums, "world": dist.get_world_size(), "gnorm": float(eng.gnorm.item()),
                "perplexity": float(eng.vq_scalars[1].item()), "offsets": eng.offsets}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""world_size-2 data-parallel test on CPU (gloo): the exchange step of the VQ-VAE train iteration.

The compute on each rank is played by the CPU oracle (the HIP kernels need a GPU); what is under test is the product's
host-side DP logic: comm-buffer layout, ONE SUM all-reduce of [grads | cnt | dw], 1/world gradient scaling before the
global-norm clip, the EMA update from GLOBAL statistics, and replica consistency afterwards."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import g2v_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


CFG = dict(n_layers=2, dropout_prob=0.0, commitment_cost=0.25, n_pre_poses=1, conditioned=True, w_l1=5.0, w_cont=0.1,
           w_var=0.5, lr=5e-4)
D, H, K, T, B_LOCAL = 12, 16, 32, 6, 8


def _shard_inputs(rank):
    g = torch.Generator().manual_seed(1234 + rank)
    x = torch.randn(B_LOCAL, T, D, generator=g)
    masks = {"dec": (torch.rand(T - 1, B_LOCAL, D, generator=g) < 0.3).to(torch.uint8)}
    return x, masks


def _local_pass(sd, x, masks):
    """Oracle forward/backward on one shard WITHOUT applying any update: raw grads, cnt, dw, loss."""
    keys = O.vqvae_trainable_keys(sd)
    leaves = {k: sd[k].detach().clone().requires_grad_(True) for k in keys}
    work = dict(sd); work.update(leaves)
    fw = O.vqvae_forward(work, x, x, CFG, True, masks)
    loss = O.custom_loss(fw["outputs"], x, 5.0, 0.1, 0.5) + fw["loss_vq"] / 400
    gl = torch.autograd.grad(loss, [leaves[k] for k in keys], allow_unused=True)
    grads = {k: (g if g is not None else torch.zeros_like(leaves[k])) for k, g in zip(keys, gl)}
    idx = fw["idx"]
    cnt = torch.bincount(idx, minlength=K).float()
    onehot = torch.zeros(idx.numel(), K); onehot[torch.arange(idx.numel()), idx] = 1
    dw = onehot.t() @ fw["flat"]
    return grads, cnt, dw


def _apply_update(sd, grads_sum, cnt, dw, world, adam):
    """What every rank does after the all-reduce: mean gradients -> clip 5 -> Adam; EMA from the GLOBAL statistics."""
    grads = {k: g / world for k, g in grads_sum.items()}
    grads, _ = O.clip_grad_norm(grads, 5.0)
    params = {k: sd[k] for k in grads}
    O.adam_step(params, grads, adam, CFG["lr"])
    sd.update(params)
    cs = sd["vq_layer._ema_cluster_size"] * 0.85 + 0.15 * cnt
    n = cs.sum()
    cs = (cs + 1e-5) / (n + K * 1e-5) * n
    ema_w = sd["vq_layer._ema_w"] * 0.85 + 0.15 * dw
    sd["vq_layer._ema_cluster_size"], sd["vq_layer._ema_w"] = cs, ema_w
    sd["vq_layer._embedding.weight"] = ema_w / cs.unsqueeze(1)


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from gesture2vec_amd import dp
    from gesture2vec_amd.engine import trainable_layout

    layout = trainable_layout(D, H, 2)
    offsets, n_flat = dp.flat_offsets(layout)
    sd = O.init_vqvae_state(D, H, 2, K, seed=100 + rank)            # ranks start DIFFERENT on purpose ...
    dp.broadcast_state([sd[k] for k in sorted(sd)], src=0)          # ... and are made identical here
    reduce_fn = dp.GradStatsAllReduce()
    adam = {}
    for step in range(2):
        x, masks = _shard_inputs(rank)
        grads, cnt, dw = _local_pass(sd, x, masks)
        comm = dp.pack_comm(layout, offsets, n_flat, grads, cnt, dw)
        reduce_fn(comm)                                              # the ONE collective of the step
        g_sum, cnt_g, dw_g = dp.unpack_comm(layout, offsets, n_flat, comm, K, H * 2)
        assert float(cnt_g.sum()) == world * B_LOCAL
        _apply_update(sd, {k: g_sum[k].clone() for k in g_sum}, cnt_g.clone(), dw_g.clone(), world, adam)
    torch.save({k: v for k, v in sd.items()}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_dp_world2_gloo(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    s0 = torch.load(os.path.join(tmp_path, "rank0.pt"))
    s1 = torch.load(os.path.join(tmp_path, "rank1.pt"))
    # replicas are bit-identical after the step (same reduced buffer, same deterministic update)
    for k in s0:
        assert torch.equal(s0[k], s1[k]), k
    # and equal to a single-process emulation that visits the two shards in turn and sums their contributions
    sd = O.init_vqvae_state(D, H, 2, K, seed=100)
    adam = {}
    for step in range(2):
        tot_g, tot_c, tot_w = None, 0, 0
        for r in range(world):
            x, masks = _shard_inputs(r)
            g, c, w = _local_pass(sd, x, masks)
            tot_g = g if tot_g is None else {k: tot_g[k] + g[k] for k in g}
            tot_c, tot_w = tot_c + c, tot_w + w
        _apply_update(sd, tot_g, tot_c, tot_w, world, adam)
    for k in s0:
        if s0[k].dtype.is_floating_point:
            np.testing.assert_allclose(s0[k].numpy(), sd[k].numpy(), rtol=1e-5, atol=1e-6, err_msg=k)


def test_comm_layout_roundtrip():
    from gesture2vec_amd import dp
    from gesture2vec_amd.engine import trainable_layout
    layout = trainable_layout(135, 64, 2)
    offsets, n_flat = dp.flat_offsets(layout)
    assert n_flat % 4 == 0 and all(off % 4 == 0 for off, _, _ in offsets.values())
    g = {name: torch.randn(shape) for name, shape in layout}
    cnt, dw = torch.arange(512.0), torch.randn(512, 128)
    comm = dp.pack_comm(layout, offsets, n_flat, g, cnt, dw)
    g2, c2, w2 = dp.unpack_comm(layout, offsets, n_flat, comm, 512, 128)
    assert all(torch.equal(g[k], g2[k]) for k in g) and torch.equal(c2, cnt) and torch.equal(w2, dw)
    assert comm.numel() == n_flat + 512 + 512 * 128 + 4          # (+ the fault-flag slot of round 5)

"""custom_loss (train_eval/train_seq2seq.py:40-88) carried by the CHASER kernel beside the persistent forward rollout and by the
backward rollout's tile load (g2v_custom_loss_chase, g2v_dec_saved.loss_*; include/g2v.h) against the separate
g2v_custom_loss_fwd_bwd launch between the two rollouts: both form dLoss/dy through the same device functions (csrc/common.hpp:
loss_sign_code / loss_grad_const / loss_col_coef / loss_grad), so every gradient, and with it every weight, Adam moment and the
codebook, must be BITWISE equal after three fused train steps; the four loss sums are added in a different order (per workgroup
tile instead of per 256 columns), so the loss terms agree to fp32 summation accuracy."""
import pytest
import torch

from oracle import g2v_oracle as O
from test_gpu_dp_engine import _engine

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("B,p,n_pre,conditioned,T", [(64, 0.0, 1, True, 34), (48, 0.2, 1, True, 34), (32, 0.0, 4, True, 34),
                                                     (32, 0.2, 1, False, 34), (1024, 0.0, 1, True, 34), (4096, 0.0, 1, True, 34),
                                                     (32, 0.0, 1, True, 2), (32, 0.0, 1, True, 3), (16, 0.0, 1, True, 4)])
@pytest.mark.parametrize("graph", [False, True])
def test_loss_chaser_beside_the_rollout_equals_the_separate_loss_kernel(B, p, n_pre, conditioned, T, graph):
    """eager launches and the step replayed from a hipGraph (the chaser is a parallel branch of the graph); small batches run
    the branch regime through overlap_min_rows = 0 (one row of workgroups: the exchange's one-hop form; teacher forcing, the
    unconditioned decoder, inter-layer dropout, T = 2 .. 4: no / one / two chased steps)."""
    if graph and B not in (64, 1024, 4096):
        pytest.skip("graph replay is covered at three sizes")
    D, H, K = 135, 64, 512
    sd = O.init_vqvae_state(D, H, 2, K, seed=11)
    kw = dict(lr=5e-4, w_l1=5.0, w_cont=0.1, w_var=0.5)
    xs = [torch.randn(B, T, D, generator=torch.Generator().manual_seed(170 + s)).to(DEV) for s in range(3)]
    xs[1][:, :, 7] = xs[1][:, :1, 7]       # a column the model cannot match exactly but whose target is constant in time
    got = {}
    for chase in (False, True):
        eng = _engine(sd, D, H, K, T, p)
        eng.seed = 5
        eng.n_pre, eng.conditioned = n_pre, conditioned
        eng.loss_chase = chase
        eng.overlap_min_rows = 0
        terms, dys = [], []
        xg = torch.empty_like(xs[0])
        g = None
        for s, x in enumerate(xs):
            xg.copy_(x)
            if graph and s == 1:             # step 0 ran eagerly (it sizes the workspaces); capture, then replay steps 1 and 2
                g = torch.cuda.CUDAGraph()
                torch.cuda.synchronize()
                with torch.cuda.graph(g):
                    eng.train_step(xg, xg, **kw)
            if g is not None:
                g.replay()
            else:
                eng.train_step(xg, xg, **kw)
            assert eng.buffers(B)["loss_folded"] is chase          # the path under test is the one that ran
            terms.append(eng.loss_terms.clone())
            dys.append(eng.buffers(B)["dy"][1:].clone())
        torch.cuda.synchronize()
        eng.check_faults()
        got[chase] = (eng, terms, dys)
    (ref, t_ref, dy_ref), (eng, t_got, dy_got) = got[False], got[True]
    for s in range(3):
        assert torch.equal(dy_got[s], dy_ref[s]), ("dy", s)
        torch.testing.assert_close(t_got[s], t_ref[s], rtol=2e-5, atol=1e-7)
    for name in ("flat", "m", "v", "codebook", "ema_w", "ema_cs", "bn_rm", "bn_rv", "vq_scalars"):
        assert torch.equal(getattr(eng, name), getattr(ref, name)), name


def test_loss_fields_are_refused_where_the_rollout_cannot_carry_the_loss():
    """B = 40 is not a multiple of 16: the per-step kernels run, g2v_dec_rollout_fuses_loss says 0, the engine keeps the
    separate loss kernel -- and the C entry point refuses the loss_* fields instead of ignoring them."""
    import ctypes as C
    from gesture2vec_amd._lib import check
    T, D, H, K, B = 34, 135, 64, 512, 40
    sd = O.init_vqvae_state(D, H, 2, K, seed=3)
    eng = _engine(sd, D, H, K, T, 0.0)
    assert eng.lib.g2v_dec_rollout_fuses_loss(B, D, H, T) == 0
    assert eng.lib.g2v_dec_rollout_fuses_loss(64, D, H, T) == 1
    x = torch.randn(B, T, D, generator=torch.Generator().manual_seed(1)).to(DEV)
    eng.train_step(x, x, lr=5e-4, w_l1=5.0, w_cont=0.1, w_var=0.5)
    b = eng.buffers(B)
    assert b["loss_folded"] is False
    ws = b["ws"]
    rc = eng.lib.g2v_dec_rollout_fwd(x.data_ptr(), b["quant"].data_ptr(), C.byref(eng.dec_wstruct()), C.byref(b["sv_loss"]),
                                     b["keep95"].data_ptr(), None, 0.0, 1, 1, 1, T, B, D, H, ws.data_ptr(), ws.numel(), None)
    assert rc != 0 and b"fuses_loss" in eng.lib.g2v_last_error()
    torch.cuda.synchronize()


@pytest.mark.parametrize("B,p", [(64, 0.0), (48, 0.2), (1024, 0.0)])
def test_quantiser_backward_inside_the_bptt_kernel_changes_nothing(B, p):
    """g2v_gru_dir_bwd.hn_*: the straight-through + commitment gradient (g2v_vq_bwd, reference :1285-1292) formed by the encoder's
    BPTT kernel where it reads its incoming gradient, against the separate launch: the same fma per element, so three fused
    train steps must leave bitwise equal state (epoch 0 -- the commitment term switched off -- included)."""
    T, D, H, K = 34, 135, 64, 512
    sd = O.init_vqvae_state(D, H, 2, K, seed=21)
    kw = dict(lr=5e-4, w_l1=5.0, w_cont=0.1, w_var=0.5)
    xs = [torch.randn(B, T, D, generator=torch.Generator().manual_seed(270 + s)).to(DEV) for s in range(3)]
    got = {}
    for fused in (False, True):
        eng = _engine(sd, D, H, K, T, p)
        eng.seed = 5
        eng._fuse_vq_bwd = fused
        for s, x in enumerate(xs):
            eng.train_step(x, x, epoch=s, **kw)          # epoch 0: g_loss_vq = 0
        torch.cuda.synchronize()
        got[fused] = eng
    ref, eng = got[False], got[True]
    for name in ("flat", "m", "v", "codebook", "ema_w", "ema_cs", "bn_rm", "bn_rv", "vq_scalars", "loss_terms"):
        assert torch.equal(getattr(eng, name), getattr(ref, name)), name


def test_input_dropout_drawn_inside_its_kernel_equals_the_explicit_mask_route():
    """The fused step forms the encoder's dropped input with the mask drawn INSIDE the kernel (g2v_dropout_rows) and draws the
    other masks at counter offsets + 1 / + 2 with one advance of the counter; an engine whose masks are drawn beforehand
    (draw_masks("all"): the mask tensors, g2v_mask_rows) and which is then stepped with draw_masks=False must end bitwise equal,
    Philox counter included."""
    B, p = 64, 0.2
    T, D, H, K = 34, 135, 64, 512
    sd = O.init_vqvae_state(D, H, 2, K, seed=31)
    kw = dict(lr=5e-4, w_l1=5.0, w_cont=0.1, w_var=0.5)
    xs = [torch.randn(B, T, D, generator=torch.Generator().manual_seed(370 + s)).to(DEV) for s in range(3)]
    a, b = _engine(sd, D, H, K, T, p), _engine(sd, D, H, K, T, p)
    a.seed = b.seed = 77
    for x in xs:
        a.train_step(x, x, **kw)                           # masks drawn by the step (input mask inside its kernel)
        b.draw_masks(B, True)                              # the same stream, as tensors
        b.train_step(x, x, draw_masks=False, **kw)
    torch.cuda.synchronize()
    assert int(a.rng_counter) == int(b.rng_counter) == 9
    for name in ("flat", "m", "v", "codebook", "ema_w", "ema_cs", "bn_rm", "bn_rv", "vq_scalars", "loss_terms"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    for k in ("keep95", "keep_l0", "x_drop"):
        assert torch.equal(a.buffers(B)[k], b.buffers(B)[k]), k


@pytest.mark.parametrize("B", [256, 4096])
def test_latched_fault_reaches_train_iter_and_the_step_is_not_applied(B):
    """A latched residency fault of the persistent rollouts (include/g2v.h; injected through the test hook
    g2v_dec_rollout_persist_fault(-1)): the kernels that commit a step -- clip + Adam, the EMA codebook update, the BatchNorm
    running statistics -- leave the model state as it was; train_iter sees the latch in its one read-back
    (g2v_iteration_readback), the persistent path is switched off and the iteration is repeated in-process on the per-step
    kernels."""
    import bench
    from gesture2vec_amd import _lib
    from gesture2vec_amd.model.Autoencoder_VQVAE_model import Autoencoder_VQVAE
    from gesture2vec_amd.train_eval.train_seq2seq import FusedClipAdam, train_iter_Autoencoder_VQ_seq2seq
    lib = _lib.load()
    args = bench.model_args()
    args.loss_l1_weight, args.loss_cont_weight, args.loss_var_weight, args.learning_rate = 5.0, 0.1, 0.5, 5e-4
    T, D = 34, 135
    torch.manual_seed(5)
    net = Autoencoder_VQVAE(args, D, T).to(DEV)
    net.train(True)
    optim = FusedClipAdam(net, lr=5e-4, betas=(0.5, 0.999))
    x = torch.randn(B, T, D, generator=torch.Generator().manual_seed(2)).to(DEV)
    try:
        for _ in range(3):                                  # eager, capture, replay
            loss, _ = train_iter_Autoencoder_VQ_seq2seq(args, 1, x, x, net, optim)
        eng = net.engine()
        snap = [t.clone() for t in (eng.flat, eng.codebook, eng.ema_w, eng.ema_cs, eng.bn_rm, eng.bn_rv, eng.step_counter)]
        assert lib.g2v_dec_rollout_persist_fault(-1) == 1
        # the engine alone: the faulted step is not applied
        eng.train_step(x, x, lr=5e-4, w_l1=5.0, w_cont=0.1, w_var=0.5)
        torch.cuda.synchronize()
        nbt = int(net.decoder.decoder.pre_linear[1].num_batches_tracked)
        for was, now in zip(snap, (eng.flat, eng.codebook, eng.ema_w, eng.ema_cs, eng.bn_rm, eng.bn_rv, eng.step_counter)):
            assert torch.equal(was, now), "a faulted step must not be applied"
        # train_iter (round 5): sees the latch in its read-back, logs a warning and REPEATS the iteration in-process on the
        # per-step kernels -- one valid step is applied, the training run goes on
        assert lib.g2v_dec_rollout_persist_fault(0) == 1
        loss2, _ = train_iter_Autoencoder_VQ_seq2seq(args, 1, x, x, net, optim)
        assert lib.g2v_dec_rollout_persist_fault(0) == 0    # check_faults() cleared the latch ...
        assert eng.ctx.get(_lib.OPT_PERSISTENT) == 0        # ... and switched the persistent path off IN THE ENGINE'S OWN CONTEXT
        assert lib.g2v_ctx_get_option(None, _lib.OPT_PERSISTENT) == 1     # (the process's default context is untouched: round 6)
        assert abs(loss2["loss"] - loss["loss"]) <= 0.05 * abs(loss["loss"]) and not torch.equal(snap[0], eng.flat)
        assert int(eng.step_counter) == int(snap[6]) + 1, "exactly one step was applied"
        assert int(net.decoder.decoder.pre_linear[1].num_batches_tracked) == nbt + (T - 1)
        loss3, _ = train_iter_Autoencoder_VQ_seq2seq(args, 1, x, x, net, optim)
        assert abs(loss3["loss"] - loss["loss"]) <= 0.05 * abs(loss["loss"])
    finally:
        lib.g2v_dec_rollout_persist_fault(1)
        lib.g2v_dec_rollout_set_persistent(1)
        lib.g2v_gru_seq_set_cluster(1)


def test_latched_fault_on_the_cluster_kernels_of_the_shipped_shape_is_repeated_on_the_per_step_launches():
    """config/VQ-VAE.yml as shipped (B = 128, H = 200): encoder GRU and decoder rollout run as persistent CLUSTER kernels (round 5),
    which latch the same fault word.  The same contract as above: a faulted step is not applied, train_iter repeats the iteration
    on the per-step launches (g2v_dec_rollout_set_persistent(0) + g2v_gru_seq_set_cluster(0)) and training goes on."""
    import bench
    from gesture2vec_amd import _lib
    from gesture2vec_amd.model.Autoencoder_VQVAE_model import Autoencoder_VQVAE
    from gesture2vec_amd.train_eval.train_seq2seq import FusedClipAdam, train_iter_Autoencoder_VQ_seq2seq
    lib = _lib.load()
    saved_cfg = dict(bench.CFG)
    try:
        bench.CFG.update({k: v for k, v in bench.CONFIGS["native"].items() if k != "name"})
        args = bench.model_args()
        args.loss_l1_weight, args.loss_cont_weight, args.loss_var_weight, args.learning_rate = 5.0, 0.1, 0.5, 5e-4
        T, D, B = bench.CFG["T"], bench.CFG["D"], 128
        assert lib.g2v_dec_rollout_cluster_ok(B, D, int(args.hidden_size)) == 1
        torch.manual_seed(5)
        net = Autoencoder_VQVAE(args, D, T).to(DEV)
        net.train(True)
        optim = FusedClipAdam(net, lr=5e-4, betas=(0.5, 0.999))
        x = torch.randn(B, T, D, generator=torch.Generator().manual_seed(2)).to(DEV)
        for _ in range(3):                                  # eager, capture, replay
            loss, _ = train_iter_Autoencoder_VQ_seq2seq(args, 1, x, x, net, optim)
        eng = net.engine()
        snap = [t.clone() for t in (eng.flat, eng.codebook, eng.ema_w, eng.ema_cs, eng.bn_rm, eng.bn_rv, eng.step_counter)]
        assert lib.g2v_dec_rollout_persist_fault(-1) == 1
        eng.train_step(x, x, lr=5e-4, w_l1=5.0, w_cont=0.1, w_var=0.5)
        torch.cuda.synchronize()
        for was, now in zip(snap, (eng.flat, eng.codebook, eng.ema_w, eng.ema_cs, eng.bn_rm, eng.bn_rv, eng.step_counter)):
            assert torch.equal(was, now), "a faulted step must not be applied"
        loss2, _ = train_iter_Autoencoder_VQ_seq2seq(args, 1, x, x, net, optim)
        assert lib.g2v_dec_rollout_persist_fault(0) == 0
        assert eng.ctx.get(_lib.OPT_PERSISTENT) == 0 and eng.ctx.get(_lib.OPT_GRU_CLUSTER) == 0
        with eng.ctx:
            assert lib.g2v_dec_rollout_cluster_ok(B, D, int(args.hidden_size)) == 0
        assert lib.g2v_dec_rollout_cluster_ok(B, D, int(args.hidden_size)) == 1     # (another engine, the default context: still offered)
        assert abs(loss2["loss"] - loss["loss"]) <= 0.05 * abs(loss["loss"]) and not torch.equal(snap[0], eng.flat)
        assert int(eng.step_counter) == int(snap[6]) + 1, "exactly one step was applied"
        loss3, _ = train_iter_Autoencoder_VQ_seq2seq(args, 1, x, x, net, optim)
        assert abs(loss3["loss"] - loss["loss"]) <= 0.05 * abs(loss["loss"])
    finally:
        bench.CFG.clear()
        bench.CFG.update(saved_cfg)
        lib.g2v_dec_rollout_persist_fault(1)
        lib.g2v_dec_rollout_set_persistent(1)
        lib.g2v_gru_seq_set_cluster(1)


@pytest.mark.parametrize("B", [256, 4096])
def test_fault_latched_in_the_middle_of_a_step_leaves_the_whole_model_state_untouched(B):
    """Round 5 (advisor finding): a fault that appears AFTER the forward rollout -- in the backward rollout, or the chaser's latch 2 --
    used to find the EMA codebook update already committed (it ran beside the first steps of the forward rollout) and BatchNorm's
    running statistics written (end of the forward rollout).  Both commits now sit behind the backward rollout, gated on the
    latch like clip + Adam: latch injected between the forward and the backward of ONE fused step (eager launches, host sync in
    between) -> parameters, Adam moments, step counter, codebook, EMA statistics and BatchNorm running statistics are bitwise
    what they were; the same step without the injection changes all of them."""
    from gesture2vec_amd import _lib
    lib = _lib.load()
    D, H, K, T, p = 135, 64, 512, 34, 0.0
    sd = O.init_vqvae_state(D, H, 2, K, seed=3)
    eng = _engine(sd, D, H, K, T, p)
    x = torch.randn(B, T, D, generator=torch.Generator().manual_seed(11)).to(DEV)
    kw = dict(lr=5e-4, w_l1=5.0, w_cont=0.1, w_var=0.5)
    names = ("flat", "m", "v", "step_counter", "codebook", "ema_w", "ema_cs", "bn_rm", "bn_rv")
    try:
        eng.train_step(x, x, **kw)                          # a first, ordinary step (EMA statistics no longer at their init)
        torch.cuda.synchronize()
        snap = {n: getattr(eng, n).clone() for n in names}
        real_loss = eng.loss
        def loss_then_fault(*a, **k):                       # between the forward (rollout + chaser joined by backward) and the backward
            real_loss(*a, **k)
            torch.cuda.synchronize()
            assert lib.g2v_dec_rollout_persist_fault(-2) == 2
        eng.loss = loss_then_fault
        eng.train_step(x, x, **kw)
        eng.loss = real_loss
        torch.cuda.synchronize()
        for n in names:
            assert torch.equal(snap[n], getattr(eng, n)), f"{n} changed in a faulted step"
        with pytest.raises(RuntimeError, match="persistent rollout kernel"):
            eng.check_faults()
        assert lib.g2v_dec_rollout_persist_fault(0) == 0
        eng.ctx.set(_lib.OPT_PERSISTENT, 1)                 # (check_faults switched it off: this test goes on with the same kernels)
        eng.ctx.set(_lib.OPT_GRU_CLUSTER, 1)
        eng.rearm()
        eng.train_step(x, x, **kw)
        torch.cuda.synchronize()
        for n in names:
            assert not torch.equal(snap[n], getattr(eng, n)), f"{n} did not change in a valid step"
    finally:
        lib.g2v_dec_rollout_persist_fault(1)
        lib.g2v_dec_rollout_set_persistent(1)
        lib.g2v_gru_seq_set_cluster(1)


def test_a_faulting_rank_makes_every_rank_skip_the_step_under_data_parallelism():
    """Round 5 (advisor finding): the fault latch is per process, but a rank whose rollout faulted has already fed garbage gradients
    into the all-reduce.  Each rank writes its latch into the flag slot of the communication buffer in front of the SUM
    (g2v_dec_rollout_fault_flag), and a non-zero reduced flag latches every rank behind it: EMA update, BatchNorm statistics and
    clip + Adam -- all behind the all-reduce under data parallelism -- leave the state alone everywhere.  One process here:
    (i) a latched rank reports 1 in its slot, (ii) a rank with a clear latch that receives a non-zero flag skips the step."""
    from gesture2vec_amd import _lib
    lib = _lib.load()
    D, H, K, T, B = 135, 64, 512, 34, 256
    sd = O.init_vqvae_state(D, H, 2, K, seed=3)
    eng = _engine(sd, D, H, K, T, 0.0)
    x = torch.randn(B, T, D, generator=torch.Generator().manual_seed(11)).to(DEV)
    kw = dict(w_l1=5.0, w_cont=0.1, w_var=0.5)
    names = ("flat", "m", "v", "step_counter", "codebook", "ema_w", "ema_cs", "bn_rm", "bn_rv")
    try:
        eng.train_step_local(x, x, dp=True, **kw)
        eng.train_step_apply(B, lr=5e-4, world=1, dp=True)            # an ordinary data-parallel step (a world of one)
        torch.cuda.synchronize()
        assert float(eng.fault_flag[0]) == 0.0
        snap = {n: getattr(eng, n).clone() for n in names}
        # (i) this rank faults: its slot says so
        assert lib.g2v_dec_rollout_persist_fault(-1) == 1
        eng.train_step_local(x, x, dp=True, **kw)
        torch.cuda.synchronize()
        assert float(eng.fault_flag[0]) == 1.0
        assert lib.g2v_dec_rollout_persist_fault(1) == 1              # read + clear: now THIS rank is healthy ...
        eng.fault_flag[0] = 2.0                                       # ... and the reduced flag says two OTHER ranks faulted
        eng.train_step_apply(B, lr=5e-4, world=1, dp=True)
        torch.cuda.synchronize()
        assert lib.g2v_dec_rollout_persist_fault(0) == 3              # (ii) latched by the flag
        for n in names:
            assert torch.equal(snap[n], getattr(eng, n)), f"{n} changed in a step another rank poisoned"
        assert lib.g2v_dec_rollout_persist_fault(1) == 3
        eng.train_step_local(x, x, dp=True, **kw)
        eng.train_step_apply(B, lr=5e-4, world=1, dp=True)
        torch.cuda.synchronize()
        for n in names:
            assert not torch.equal(snap[n], getattr(eng, n)), f"{n} did not change in a valid step"
    finally:
        lib.g2v_dec_rollout_persist_fault(1)
        lib.g2v_dec_rollout_set_persistent(1)
        lib.g2v_gru_seq_set_cluster(1)


def test_two_engines_do_not_share_switches_and_a_fault_in_one_leaves_the_other_on_the_fast_path():
    """Round 6 (round-5 verdict, boundary): the library's implementation switches live in a caller-owned context (include/g2v.h:
    g2v_ctx) that each engine binds around its calls.  Engine A latches a fault and falls back to the per-step launches; engine B,
    in the same process, keeps the persistent kernels; A's policy re-arms A's fast path after its clean interval."""
    from gesture2vec_amd import _lib
    lib = _lib.load()
    D, H, K, T, B = 135, 64, 512, 34, 256
    sd = O.init_vqvae_state(D, H, 2, K, seed=3)
    ea, eb = _engine(sd, D, H, K, T, 0.0), _engine(sd, D, H, K, T, 0.0)
    x = torch.randn(B, T, D, generator=torch.Generator().manual_seed(11)).to(DEV)
    kw = dict(lr=5e-4, w_l1=5.0, w_cont=0.1, w_var=0.5)
    try:
        assert lib.g2v_dec_rollout_persist_fault(1) in (0, 1, 2, 3)
        ea.train_step(x, x, **kw); eb.train_step(x, x, **kw)
        torch.cuda.synchronize()
        with ea.ctx:
            tiles_a = lib.g2v_dec_rollout_tiles_per_workgroup(B, D, H)
        assert tiles_a >= 1, "this shape is expected on the persistent rollouts"
        assert lib.g2v_dec_rollout_persist_fault(-1) == 1
        ea.train_step(x, x, **kw)
        torch.cuda.synchronize()
        with pytest.raises(RuntimeError, match="persistent rollout kernel"):
            ea.check_faults()
        assert ea.ctx.get(_lib.OPT_PERSISTENT) == 0 and ea.fault_policy.off
        assert eb.ctx.get(_lib.OPT_PERSISTENT) == 1 and not eb.fault_policy.off
        with ea.ctx:
            assert lib.g2v_dec_rollout_tiles_per_workgroup(B, D, H) == 0
        with eb.ctx:
            assert lib.g2v_dec_rollout_tiles_per_workgroup(B, D, H) == tiles_a
        # both keep training: A on the per-step launches, B on the persistent ones -- same arithmetic to rounding
        ea.train_step(x, x, **kw); eb.train_step(x, x, **kw)
        torch.cuda.synchronize()
        assert lib.g2v_dec_rollout_persist_fault(0) == 0
        # A's policy brings A's fast path back after its clean interval
        ea.fault_policy.rearm_after = 3
        for _ in range(3):
            ea.train_step(x, x, **kw)
            if ea.fault_policy.tick():
                ea.rearm()
        assert ea.ctx.get(_lib.OPT_PERSISTENT) == 1 and not ea.fault_policy.off and ea.fault_policy.rearms == 1
        ea.train_step(x, x, **kw)
        torch.cuda.synchronize()
        assert lib.g2v_dec_rollout_persist_fault(0) == 0
    finally:
        lib.g2v_dec_rollout_persist_fault(1)

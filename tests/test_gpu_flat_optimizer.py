"""FlatClipAdam (one-launch gradient gather + fused clip / Adam over a flat buffer) against torch's own
clip_grad_norm_ + Adam on a parameter LIST with grad-less tensors in it (what the reference's train_iter functions run,
train_eval/train_seq2seq.py:540-546, 743-744), and the error behaviour of the entry points added in round 2."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_flat_clip_adam_matches_torch_with_gradless_tensors():
    from gesture2vec_amd.flat import FlatClipAdam
    g = torch.Generator().manual_seed(4)
    shapes = [(7, 5), (13,), (3, 4, 2), (64, 33), (5,), (9, 9), (1,), (6, 3)]
    mine = [torch.nn.Parameter(torch.randn(*s, generator=g).to(DEV)) for s in shapes]
    ref = [torch.nn.Parameter(p.detach().clone().cpu().double()) for p in mine]
    opt = FlatClipAdam(mine, lr=1e-2, betas=(0.5, 0.999), max_norm=0.7)
    ropt = torch.optim.Adam(ref, lr=1e-2, betas=(0.5, 0.999))
    # step -> tensors without a gradient (2, 3 adjacent: one merged range).  A tensor that drops out stays out: one that came
    # back would see the GLOBAL step count in its bias correction here and its own in torch (documented in flat.py)
    # Tensor 7 NEVER receives a gradient (Part d without attention: the encoder's second layer): its moments stay exactly zero,
    # so the fused step leaves it bitwise alone without the keep / restore copies (round 6).
    gradless = {0: (7,), 1: (2, 3, 7), 2: (2, 3, 5, 7), 3: (0, 2, 3, 5, 7)}
    never = mine[7].detach().clone()
    for step in range(4):
        opt.zero_grad()
        ropt.zero_grad(set_to_none=True)
        for k, (p, r) in enumerate(zip(mine, ref)):
            if k in gradless[step]:
                continue
            gr = torch.randn(*shapes[k], generator=g) * (3.0 if step % 2 else 0.05)     # clipped and unclipped steps
            if k == 4:
                gr = gr.t().contiguous().t() if gr.dim() == 2 else gr
            p.grad = gr.to(DEV)
            r.grad = gr.double()
        opt.step()
        torch.nn.utils.clip_grad_norm_([r for r in ref if r.grad is not None], 0.7)
        ropt.step()
        for k, (p, r) in enumerate(zip(mine, ref)):
            err = float((p.detach().cpu().double() - r.detach()).abs().max())
            assert err <= 2e-6 * max(1.0, float(r.detach().abs().max())), (step, k, err)
    assert torch.equal(mine[7].detach(), never)
    o7 = opt.fp.offsets[7]
    assert float(opt.fp.m[o7:o7 + 18].abs().max()) == 0.0 and float(opt.fp.v[o7:o7 + 18].abs().max()) == 0.0
    assert all(o != o7 for o, _ in opt.fp.skipped)                  # (no keep / restore range for it)
    # a tensor that skipped a step was left exactly where it was (no moment decay, no stale-momentum move)
    before = mine[6].detach().clone()
    opt.zero_grad()
    for k, p in enumerate(mine):
        if k not in (6, 7):
            p.grad = torch.ones_like(p)
    opt.step()
    assert torch.equal(mine[6].detach(), before)
    assert torch.equal(mine[7].detach(), never)


def test_copy_segments_zero_fill_and_many_segments():
    from gesture2vec_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    n = [int(v) for v in torch.randint(1, 300, (70,), generator=g)]          # 70 segments: two launches of <= 48
    srcs = [torch.randn(k, generator=g).to(DEV) if i % 7 else None for i, k in enumerate(n)]
    dst = torch.full((sum(n) + 3,), 9.0, device=DEV)
    offs, o = [], 0
    for k in n:
        offs.append(o)
        o += k
    sp = (C.c_void_p * len(n))(*[s.data_ptr() if s is not None else None for s in srcs])
    dp = (C.c_void_p * len(n))(*[dst.data_ptr() + 4 * o_ for o_ in offs])
    nn_ = (C.c_int64 * len(n))(*n)
    assert lib.g2v_copy_segments(sp, dp, nn_, len(n), torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    for s, o_, k in zip(srcs, offs, n):
        want = s if s is not None else torch.zeros(k, device=DEV)
        assert torch.equal(dst[o_:o_ + k], want)
    assert float(dst[-3:].min()) == 9.0


def test_round2_entry_points_refuse_bad_arguments():
    from gesture2vec_amd import _lib, ops
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    x = torch.zeros(8, 50, device=DEV)           # in_dim 50: not a multiple of 4
    h = torch.zeros(8, 48, device=DEV)
    wi, wh = torch.zeros(144, 50, device=DEV), torch.zeros(144, 48, device=DEV)
    b = torch.zeros(144, device=DEV)
    with pytest.raises(RuntimeError, match="multiples of 4"):
        ops.gru_cell_fwd(x, h, wi, wh, b, b)
    flat, W = torch.zeros(64, 100, device=DEV), torch.zeros(512, 100, device=DEV)      # E = 100: the bulk path is E == 128 only
    with pytest.raises(RuntimeError, match="E == 128"):
        ops.vq_assign_bulk(flat, W, torch.zeros(512, device=DEV))
    flat, W = torch.zeros(64, 128, device=DEV), torch.zeros(512, 128, device=DEV)
    idx = torch.zeros(64, dtype=torch.int64, device=DEV)
    ws = torch.zeros(16, dtype=torch.uint8, device=DEV)
    rc = lib.g2v_vq_assign_bulk(flat.data_ptr(), W.data_ptr(), torch.zeros(512, device=DEV).data_ptr(), idx.data_ptr(), 64, 128, 512,
                                ws.data_ptr(), ws.numel(), None, st)
    assert rc != 0 and b"workspace" in lib.g2v_last_error()
    rc = lib.g2v_embedding_bwd(flat.data_ptr(), idx.data_ptr(), None, 1.0, W.data_ptr(), 64, 128, 512, 1, ws.data_ptr(), 16, st)
    assert rc != 0 and b"workspace" in lib.g2v_last_error()          # even the one-launch path wants its 256-byte minimum
    rc = lib.g2v_gru_seq_prepare(None, None, 2, 64, 1, ws.data_ptr(), 16, None, 0, st)
    assert rc != 0 and b"null" in lib.g2v_last_error()

#!/usr/bin/env python3
"""bench.py -- gesture-chunks/sec of one full VQ-VAE train iteration (BASELINE.json metric).

One "step" = one `train_iter_Autoencoder_VQ_seq2seq` equivalent on a batch of synthetic pose chunks already
resident in HBM: Philox dropout masks -> encoder -> EMA quantiser (assign + stats + EMA update) -> T-1 step decoder
rollout -> custom_loss -> full backward -> [DP: one RCCL all-reduce of grads + EMA stats] -> fused clip+Adam.
Workload at N=1: BASELINE.json configs[1] "VQ-VAE.yml full" = B=4096, T=34, D=135, H=64, L=2 (E=128), K=512, fp32.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      the VQ assign kernel (the kernel BASELINE.json's north star names), timed live with events on the
                launch stream; fp32 MFMA bound (arithmetic intensity K/4 flop/B > fp32 ridge)
  cpu_baseline  the CPU oracle (oracle/g2v_oracle.py, kind "port") timed on this host's cores on a bounded sample
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CFG = dict(B=4096, T=34, D=135, H=64, L=2, K=512, beta=0.25, dropout_prob=0.0, lr=5e-4, w_l1=5.0, w_cont=0.1, w_var=0.5)
# --config: the workload.  "full" is the contract line (BASELINE.json configs[1]); the others put the shapes the reference's own
# YAMLs ship on the same line format, e.g. for the driver's multi-GPU runs of configs[4]:
#   python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N --config genea
CONFIGS = {
    "full": dict(B=4096, T=34, D=135, H=64, K=512, dropout_prob=0.0, lr=5e-4,
                 name="BASELINE configs[1] (VQ-VAE full shape)"),
    "native": dict(B=128, T=20, D=40, H=200, K=512, dropout_prob=0.2, lr=5e-4,
                   name="config/VQ-VAE.yml as shipped (n_poses 20, rep_learning_dim 40 = the DAE-stacked input of BASELINE configs[2], "
                        "hidden_size 200, K 512, batch_size 128, dropout_prob 0.2)"),
    "genea": dict(B=4096, T=10, D=45, H=200, K=400, dropout_prob=0.0, lr=1e-4,
                  name="BASELINE configs[4] (config/VQ-VAE_GENEA.yml shape: n_poses 10, rep_learning_dim 45, hidden_size 200, K 400, "
                       "lr 1e-4; B_local 4096 per GPU, weak scaling)"),
}
PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
PEAK_HBM_GBS = 8000.0


def model_args():
    return argparse.Namespace(
        rep_learning_dim=CFG["D"], hidden_size=CFG["H"], n_layers=CFG["L"], dropout_prob=CFG["dropout_prob"],
        autoencoder_vae="False", autoencoder_vq="True", autoencoder_vq_components=CFG["K"],
        autoencoder_vq_commitment_cost=CFG["beta"], n_pre_poses=1, autoencoder_conditioned="True",
        autoencoder_att="False", autoencoder_fixed_weight="False", n_poses=CFG["T"])


def cpu_baseline(B_main: int):
    """Oracle (CPU restatement, parity-pinned to the reference) on the host cores, SURVEY.md 8(d): the same train step at
    the benchmark's own batch (B = 4096) and at the reference's native batch (B = 128); MEDIAN of 10 steps each at the
    calibrated thread count (8 / 16 / 32, whichever is fastest at B = 128: on a 256-core host the many tiny ops of the
    T-1 step Python loop get slower, not faster, with every core in the pool), plus 3 steps at B = 4096 with every host core
    (torch.set_num_threads(os.cpu_count())), CPU model printed.  About a minute of CPU work in all."""
    import statistics
    from oracle import g2v_oracle as O
    T, D, H, K = CFG["T"], CFG["D"], CFG["H"], CFG["K"]
    cfg = dict(n_layers=2, dropout_prob=CFG["dropout_prob"], commitment_cost=CFG["beta"], n_pre_poses=1, conditioned=True,
               w_l1=CFG["w_l1"], w_cont=CFG["w_cont"], w_var=CFG["w_var"], lr=CFG["lr"])
    p = CFG["dropout_prob"]

    def inputs(Bs):
        g = torch.Generator().manual_seed(1234)
        x = torch.randn(Bs, T, D, generator=g)
        masks = {"dec": (torch.rand(T - 1, Bs, D, generator=g) < 0.05).to(torch.uint8)}
        if p > 0:
            masks["in"] = (torch.rand(T, Bs, D, generator=g) < 1 - p).to(torch.uint8)
            masks["enc_l0"] = (torch.rand(T, Bs, 2 * H, generator=g) < 1 - p).to(torch.uint8)
            masks["dec_l0"] = (torch.rand(T - 1, Bs, H, generator=g) < 1 - p).to(torch.uint8)
        return x, masks

    def one(sd, adam, x, masks):
        t0 = time.perf_counter()
        O.vqvae_train_step(sd, adam, x, masks, cfg)
        return time.perf_counter() - t0

    def timed(Bs, n_steps, budget):
        x, masks = inputs(Bs)
        sd, adam = O.init_vqvae_state(D, H, 2, K, seed=0), {}
        one(sd, adam, x, masks)             # warm-up
        ts = []
        while len(ts) < n_steps and (sum(ts) < budget or len(ts) < 3):
            ts.append(one(sd, adam, x, masks))
        return ts

    ncpu = os.cpu_count() or 1
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    best_thr, best_t = 1, float("inf")
    xs, ms = inputs(128)
    for thr in sorted({min(ncpu, c) for c in (8, 16, 32)}):
        torch.set_num_threads(thr)
        sd, adam = O.init_vqvae_state(D, H, 2, K, seed=0), {}
        one(sd, adam, xs, ms)               # warm-up at this thread count
        t = one(sd, adam, xs, ms)
        if t < best_t:
            best_thr, best_t = thr, t
        if t > 10.0:                        # pathological host: stop calibrating
            break
    torch.set_num_threads(best_thr)
    t_small = timed(128, 10, 10.0)
    t_main = timed(B_main, 10, 60.0) if B_main != 128 else t_small
    # every host core: in a child process under a hard wall-clock limit (on the pool's 256-core hosts the oracle's many small ops
    # take minutes per step with 256 threads -- the first version of this leg ran into the driver's time limit)
    t_all, all_note = [], ""
    if ncpu != best_thr:
        import subprocess
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", str(ncpu), str(B_main), "3",
                                CFG.get("config", "full")], capture_output=True, text=True, timeout=75)
            t_all = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else []
            if not t_all:
                all_note = "the child process failed"
        except subprocess.TimeoutExpired:
            all_note = f"1 warm-up + 3 steps did not finish within 75 s with {ncpu} threads"
        except Exception as e:
            all_note = f"{type(e).__name__}: {e}"
    else:
        t_all = t_main
    # second leg: the same step on torch.nn.GRU modules (ATen's fused CPU RNN kernels, the way the reference builds its model):
    # oracle/g2v_oracle_nn.py, pinned to the functional oracle in tests/test_oracle_golden.py
    fused = None
    try:
        from oracle import g2v_oracle_nn as ONN
        x, masks = inputs(B_main)
        m, opt, vq_sd = ONN.make(O.init_vqvae_state(D, H, 2, K, seed=0), D, H, 2, cfg)
        # its own thread count: this leg is a few large ATen kernels per step, which scale further than the functional oracle's
        # Python loop of small ops (round-5 verdict: "8 of 256 host cores") -- one timed step per candidate at the benchmark batch
        fthr, ft = best_thr, float("inf")
        for thr in sorted({min(ncpu, c) for c in (best_thr, 32, 64, 128)}):
            torch.set_num_threads(thr)
            ONN.train_step(m, opt, vq_sd, x, masks["dec"], cfg)          # warm-up at this thread count
            t0 = time.perf_counter()
            ONN.train_step(m, opt, vq_sd, x, masks["dec"], cfg)
            t = time.perf_counter() - t0
            if t < ft:
                fthr, ft = thr, t
            if t > 8.0:
                break
        torch.set_num_threads(fthr)
        ONN.train_step(m, opt, vq_sd, x, masks["dec"], cfg)
        tf = []
        while len(tf) < 10 and (sum(tf) < 20.0 or len(tf) < 3):
            t0 = time.perf_counter()
            ONN.train_step(m, opt, vq_sd, x, masks["dec"], cfg)
            tf.append(time.perf_counter() - t0)
        torch.set_num_threads(best_thr)
        fused = {"value": round(B_main / statistics.median(tf), 1), "unit": "chunks/s", "threads": fthr,
                 "sample": f"median of {len(tf)} steps at B={B_main}, oracle/g2v_oracle_nn.py: torch.nn.GRU modules (fused CPU RNN), "
                           "all L encoder layers executed as the reference does"}
    except Exception as e:          # a baseline leg never costs the bench line
        fused = {"value": None, "error": f"{type(e).__name__}: {e}"[:200]}
    med = statistics.median(t_main)
    functional = {"value": round(B_main / med, 1), "unit": "chunks/s", "threads": best_thr,
                  "sample": f"median of {len(t_main)} full train steps at B={B_main} (the GPU batch), {sum(t_main):.1f}s after warm-up; "
                            f"oracle/g2v_oracle.py (explicit per-step formulas: the parity checker), torch-CPU fp32"}
    # Headline = the restatement closest to the reference's own CPU path (nn.GRU modules -> ATen's fused CPU RNN kernels); the
    # Python-loop functional oracle travels as a sub-field (round-4 verdict: it is 6x slower and flatters the GPU / CPU ratio).
    head, head_sample = functional["value"], functional["sample"]
    if fused and fused.get("value"):
        head, head_sample = fused["value"], fused["sample"]
    head_thr = fused["threads"] if (fused and fused.get("value")) else best_thr
    return {"value": head, "unit": "chunks/s", "cores": head_thr, "kind": "port", "cpu_model": model,
            "fused_rnn": fused, "functional_oracle": functional,
            "host_cores": ncpu,
            "sample": head_sample + f"; {head_thr} threads (calibrated) of {ncpu} host cores ({model})",
            "all_host_cores": {"threads": ncpu, "value": (round(B_main / statistics.median(t_all), 1) if t_all else None), "unit": "chunks/s",
                               "sample": (f"median of {len(t_all)} steps at B={B_main} with torch.set_num_threads({ncpu})" if t_all else all_note)},
            "native_batch": {"B": 128, "value": round(128 / statistics.median(t_small), 1), "unit": "chunks/s",
                             "sample": f"median of {len(t_small)} steps, {best_thr} threads"},
            "note": "`value` = `fused_rnn`: the train step on torch.nn.GRU modules like the reference's own (1,947 chunks/s at survey "
                    "time on 8 threads of the build container, BASELINE.md section 2); `functional_oracle` is the parity checker "
                    "(explicit per-step formulas, a Python loop over time); reported baselines, not targets"}


def _profile_order(path: str):
    """sort key of a committed profile `rNN_<tag>_...`: round number, then the tag in the order the tags were handed out
    (a..z, aa..az, ba.. -- shorter first, then alphabetical), so `r05_ba` is newer than `r05_az` is newer than `r05_z`"""
    import re
    m = re.match(r"r(\d+)_([a-z]+)_", os.path.basename(path))
    if not m:
        return (-1, 0, "")
    return (int(m.group(1)), len(m.group(2)), m.group(2))


def pmc_traffic(kernel: str, N: int):
    """HBM bytes per launch of `kernel` at N rows, from the committed PMC summaries (separate rocprofv3 --pmc passes,
    FETCH_SIZE doubled per the gfx950 correction): profiles/*pmc_traffic.json, newest round first.  None when no
    summary holds this kernel at this size -- never a stale literal."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")), key=_profile_order, reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
        except Exception:
            continue
        e = d.get(f"N={N}")
        if e and kernel in e.get("kernel", ""):
            return int(e["hbm_bytes_per_launch_corrected"])
    return None


def in_graph_us(kernel: str):
    """duration of `kernel` INSIDE the replayed step, from the newest committed one-step timeline (profiles/*step_timeline.txt,
    rocprofv3 --kernel-trace of this command): there the kernel starts while the branch beside the encoder still holds CUs"""
    import glob
    import re
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_step_timeline.txt")), key=_profile_order, reverse=True):
        try:
            for line in open(path):
                m = re.match(r"\s*[-0-9.]+ dur\s+([0-9.]+) gap", line)
                if m and kernel in line:
                    return {"us": float(m.group(1)), "source": os.path.relpath(path, ROOT), "archived": True,
                            "note": "NOT measured by this run: read from the committed rocprofv3 timeline named in `source`"}
        except OSError:
            continue
    return None


def vq_kernel_roofline(eng, B, reps: int = 200):
    """Average duration of the VQ kernel of the product path at the benchmark size, events on the launch stream: the fused
    pre_linear + assign kernel (K6+K1+K2+K5: projection, -2 z E^T distances, argmin, gather + straight-through + SSE) on the
    engine's CURRENT state (codebook after the timed training steps).  Product path (round 3): pre_linear in fp32 MFMA, the
    distance contraction SCREENED on the bf16 matrix pipe and every code inside the error margin re-evaluated in exact fp32 in
    the same launch (results bitwise those of the fp32 kernel, tests/test_gpu_ops.py); the fp32 kernel is timed beside it."""
    from gesture2vec_amd._lib import check
    lib = eng.lib
    b = eng.buffers(B)
    N, E, K = (2 * B * eng.H) // eng.E, eng.E, eng.K
    st = torch.cuda.current_stream()
    fused = (E == 128 and K % 128 == 0)
    flops = 2.0 * N * K * E + (2.0 * N * E * E if fused else 0.0)     # 2KE (distances) + 2E^2 (pre_linear) per row
    # read z (4E) + write flat (4E) + write quantized (4E) + write idx (8, int64) per row; W_pre, b_pre, codebook, norms once
    bytes_alg = N * (12 * E + 8) + 4 * K * E + 4 * K + ((4 * E * E + 4 * E) if fused else 0)

    def timed(fn, args):
        for _ in range(20):
            check(fn(*args))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(reps):
            check(fn(*args))
        e1.record(st)
        e1.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps

    extra = {}
    eng.vq_derive()                         # operand images of the current codebook
    if fused and eng._vq_bx:
        diag = torch.zeros(4, dtype=torch.int32, device=b["flat"].device)
        mk = lambda flags, dg: (b["enc_hidden"].data_ptr(), eng.vq_wpre_frag.data_ptr(), eng.vq_pre_b.data_ptr(),
                                eng.codebook.data_ptr(), eng.vq_bx_image.data_ptr(), eng.code_sqnorm.data_ptr(),
                                b["flat"].data_ptr(), b["idx"].data_ptr(), b["quant"].data_ptr(), b["sse"].data_ptr(), dg,
                                N, E, K, flags, st.cuda_stream)
        fn, kernel = lib.g2v_vq_fused_assign_bx_fwd, "vq_fused_bx_kernel"
        us = timed(fn, mk(eng.vq_bx_flags, None))
        check(fn(*mk(eng.vq_bx_flags, diag.data_ptr())))
        idx_s = b["idx"].clone()
        us_exact = timed(fn, mk(1, None))                       # the same launch with every tile on the exact fp32 sweep
        torch.cuda.synchronize()
        d = diag.cpu().tolist()
        tiles = (N + 15) // 16
        extra = {"screening": {"tiles": tiles, "tiles_on_exact_sweep": d[0], "pairs_re_evaluated_per_tile": round(d[1] / max(tiles - d[0], 1), 2),
                               "idx_equal_to_exact_fp32_sweep_on_every_row": bool(torch.equal(idx_s, b["idx"])),
                               "avg_us_exact_fp32_sweep_same_kernel": round(us_exact, 3)},
                 "pipes": "pre_linear + candidate re-evaluation: fp32 (v_mfma_f32_16x16x4_f32 / v_fma_f32 chains); distance "
                          "screening: bf16 MFMA (v_mfma_f32_16x16x32_bf16); flops counted are the ALGORITHMIC 2NKE + 2NE^2"}
        # the round-2 fp32 kernel on the same inputs (A/B)
        frag = torch.empty(K * E, device=b["flat"].device)
        check(lib.g2v_vq_pack_codebook(eng.codebook.data_ptr(), frag.data_ptr(), K, E, st.cuda_stream))
        a_old = (b["enc_hidden"].data_ptr(), eng.vq_pre_w.data_ptr(), eng.vq_pre_b.data_ptr(), eng.codebook.data_ptr(), frag.data_ptr(),
                 eng.code_sqnorm.data_ptr(), b["flat"].data_ptr(), b["idx"].data_ptr(), b["quant"].data_ptr(), b["sse"].data_ptr(),
                 N, E, K, st.cuda_stream)
        extra["fp32_kernel_round2"] = {"kernel": "vq_fused_assign_kernel<128, true>", "avg_us": round(timed(lib.g2v_vq_fused_assign_packed_fwd, a_old), 3),
                                       "idx_equal_on_every_row": bool(torch.equal(idx_s, b["idx"]))}
        extra["fp32_kernel_round2"]["frac"] = round(flops / (extra["fp32_kernel_round2"]["avg_us"] * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)
    elif fused:
        args = (b["enc_hidden"].data_ptr(), eng.vq_pre_w.data_ptr(), eng.vq_pre_b.data_ptr(), eng.codebook.data_ptr(),
                eng.codebook_frag.data_ptr(), eng.code_sqnorm.data_ptr(), b["flat"].data_ptr(), b["idx"].data_ptr(),
                b["quant"].data_ptr(), b["sse"].data_ptr(), N, E, K, st.cuda_stream)
        fn, kernel = lib.g2v_vq_fused_assign_packed_fwd, "vq_fused_assign_kernel<128, true>"
        us = timed(fn, args)
    else:
        args = (b["flat"].data_ptr(), b["enc_hidden"].data_ptr(), eng.codebook.data_ptr(), eng.code_sqnorm.data_ptr(),
                b["idx"].data_ptr(), b["quant"].data_ptr(), None, b["sse"].data_ptr(), N, E, K, st.cuda_stream)
        fn = lib.g2v_vq_assign_fwd
        kernel = "vq_assign_rt_kernel<128, 4>" if N >= 16384 else "vq_assign_fast_kernel<128>"
        us = timed(fn, args)
    tf = flops / (us * 1e-6) / 1e12
    out = {"bound": "mfma", "achieved": round(tf, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
           "frac": round(tf / PEAK_F32_MFMA_TFLOPS, 4), "traffic": pmc_traffic(kernel, N) if (E, K) == (128, 512) else None,
           "kernel": kernel, "avg_us": round(us, 3), "flops_per_launch": flops,
           "algorithmic_bytes_per_launch": bytes_alg,
           "hbm_view_GBps": round(bytes_alg / (us * 1e-6) / 1e9, 1), "hbm_view_frac": round(bytes_alg / (us * 1e-6) / 1e9 / PEAK_HBM_GBS, 4)}
    out.update(extra)
    if (E, K, N) == (128, 512, 4096):
        out["in_step_graph"] = in_graph_us(kernel.split("<")[0])
    return out


def bulk_assign_line():
    """bulk code assignment (g2v_vq_assign_bulk: bf16 3-term split screening on the matrix pipe + exact fp32
    re-check of the undecided rows; reference call sites lmdb_data_loader.py:1274-1281, Clustering.py:151-157) at 2^20 projected
    rows, E = 128, K = 512: average of 50 calls behind 100 untimed ones (events on the launch stream), every index against the fp32 kernel's."""
    from gesture2vec_amd import ops
    N, E, K = 1 << 20, 128, 512
    g = torch.Generator().manual_seed(3)
    W = torch.randn(K, E, generator=g).to("cuda:0")
    x = torch.randn(N, E, generator=g).to("cuda:0")
    wsq = ops.vq_code_sqnorm(W)
    for _ in range(100):          # (40 ms: this leg runs behind a minute of host-side baseline work, the GPU's clocks are down)
        ops.vq_assign_bulk(x, W, wsq)
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(50):
        ops.vq_assign_bulk(x, W, wsq)
    e1.record(st)
    e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    idx, und = ops.vq_assign_bulk(x, W, wsq, want_undecided=True)
    ref = ops.vq_assign(x, None, W, wsq, want_quantized=False)[0]
    bytes_alg = N * (4 * E + 8)                              # read the row, write the index
    mfma_flops = 3 * 2.0 * N * K * E                         # the three bf16 products of the split
    return {"workload": "g2v_vq_assign_bulk, N = 2^20 rows, E = 128, K = 512, N(0,1) rows and codes", "us": round(us, 1),
            "rows_per_s": round(N / (us * 1e-6), 1), "mismatches_vs_fp32_kernel": int((idx != ref).sum()),
            "undecided_frac": round(int(und.item()) / N, 4),
            "roofline": {"bound": "mfma", "unit": "TFLOP/s", "achieved": round(mfma_flops / (us * 1e-6) / 1e12, 1), "peak": 2500.0,
                         "frac": round(mfma_flops / (us * 1e-6) / 1e12 / 2500.0, 4), "pipe": "bf16 16x16x32, executed flops of the 3-term split",
                         "hbm_view_GBps": round(bytes_alg / (us * 1e-6) / 1e9, 1),
                         "hbm_view_frac": round(bytes_alg / (us * 1e-6) / 1e9 / PEAK_HBM_GBS, 4)}}


def gru_resident_line():
    """one bidirectional GRU layer forward (g2v_gru_seq_fwd, both directions in one launch) at B = 4096, T = 20,
    H = 200 -- the encoder of the shipped YAMLs at the per-GPU batch of BASELINE configs[2] / [4] -- with W_hh resident in each CU
    (the default from 1025 rows) and streamed from L2 every step (G2V_OPT_GRU_RESIDENT_ROWS = 0); outputs compared bit for bit."""
    from gesture2vec_amd import _lib, ops
    lib = _lib.load()
    T, B, H = 20, 4096, 200
    g = torch.Generator().manual_seed(5)
    r = lambda *s: (torch.randn(*s, generator=g) * 0.3).to("cuda:0")
    base = [dict(gi=r(T, B, 3 * H), w_hh=r(3 * H, H), b_hh=r(3 * H), reverse=bool(k)) for k in range(2)]
    prev = lib.g2v_ctx_get_option(None, _lib.OPT_GRU_RESIDENT_ROWS)
    res, outs = {}, {}
    try:
        for name, rows in (("resident", 1025), ("streaming", 0)):
            lib.g2v_ctx_set_option(None, _lib.OPT_GRU_RESIDENT_ROWS, rows)
            dirs = [dict(d, h0=None, hs=torch.empty((T, B, H), device="cuda:0"), h_n=torch.empty((B, H), device="cuda:0"),
                         gates=torch.empty((T, B, 4 * H), device="cuda:0")) for d in base]
            for _ in range(20):
                ops.gru_dirs_fwd(dirs, T, B, H)
            st = torch.cuda.current_stream()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(30):
                ops.gru_dirs_fwd(dirs, T, B, H)
            e1.record(st)
            e1.synchronize()
            res[name] = e0.elapsed_time(e1) * 1e3 / 30
            outs[name] = dirs
    finally:
        lib.g2v_ctx_set_option(None, _lib.OPT_GRU_RESIDENT_ROWS, prev)
    flops = 2 * T * B * 2.0 * (3 * H) * H
    same = all(torch.equal(outs["resident"][k][n], outs["streaming"][k][n]) for k in range(2) for n in ("hs", "h_n", "gates"))
    return {"workload": "g2v_gru_seq_fwd, both directions, B = 4096, T = 20, H = 200 (pack of W_hh included)",
            "resident_us": round(res["resident"], 1), "streaming_us": round(res["streaming"], 1), "bitwise_equal": bool(same),
            "roofline": {"bound": "mfma", "unit": "TFLOP/s", "achieved": round(flops / (res["resident"] * 1e-6) / 1e12, 1),
                         "peak": PEAK_F32_MFMA_TFLOPS, "frac": round(flops / (res["resident"] * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                         "streaming_frac": round(flops / (res["streaming"] * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)}}


def part_d(with_cpu: bool):
    """"Also report text2embedding samples/s" -- config 4: n_words 3863, 300-d embeddings, lengths U{4..20} sorted
    descending, codes U{0..511} (B,6), hidden 200, 2 layers, dropout 0.2 (config/seq2seq.yml), `autoencoder_att` False and True;
    one step = train_iter_text2embedding (reference train_eval/train_seq2seq.py:462-538: forward, CE over steps 1..S-1, backward,
    clip 5, Adam) replayed from a hipGraph, B = 128 (the yml's) and 4096.  CPU: the oracle's t2e_train_step at B = 128."""
    import numpy as np
    import statistics
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from gesture2vec_amd.flat import FlatClipAdam
    from gesture2vec_amd.model.text2embedding_model import text2embedding_model
    from gesture2vec_amd.train_eval.train_seq2seq import GraphedText2EmbeddingStep
    from train_text2embedding import SyntheticSentences
    res = {"unit": "samples/s", "workload": "SURVEY.md 8(d) config 4 (config/seq2seq.yml shape: H=200, L=2, K=512, S=6, 300-d embeddings, "
                                             "3863 words), train_iter_text2embedding, hipGraph replay", "runs": []}
    keep = None
    for att in ("False", "True"):
        for B in (128, 4096):
            args = argparse.Namespace(hidden_size=200, n_layers=2, dropout_prob=0.2, autoencoder_vq_components=512, autoencoder_att=att,
                                      n_pre_poses=1, n_poses=20, sentence_frame_length=120, text2_embedding_discrete="True", batch_size=B)
            torch.manual_seed(0)
            net = text2embedding_model(args, 512, 20, 3863, 300, np.random.RandomState(0).randn(3863, 300).astype(np.float32), None).to("cuda:0")
            net.train(True)
            opt = FlatClipAdam(net.parameters(), lr=5e-4)
            data = list(SyntheticSentences(args, 3863, 1, seed=1))[0]
            ids, lengths, codes = data[0].to("cuda:0"), data[1], data[6].to("cuda:0")
            if att == "False" and B == 128:
                keep = ({k: v.detach().cpu().clone() for k, v in net.state_dict().items()}, data[0].clone(), data[1].clone(), data[6].clone())
            from gesture2vec_amd import rollout_t2e as RT
            calls0 = (RT.FUSED_CALLS, RT.CLUSTER_CALLS)
            # the sentence lengths are BAKED into the captured launches (static_lengths=True: packed positions; a trainer whose
            # lengths change per batch replays the padded-grid graph, the class's default); no latch read inside the timed
            # region (check_every=0): read_loss() below checks it once, behind the timed replays
            g = GraphedText2EmbeddingStep(args, net, opt, ids, lengths, codes, static_lengths=True, check_every=0)
            served = (RT.FUSED_CALLS - calls0[0], RT.CLUSTER_CALLS - calls0[1])      # which route the captured step took (counted, not assumed)
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize()
            n = 30
            t0 = time.perf_counter()
            for _ in range(n):
                g.replay()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            # flops as EXECUTED: forward x 3.  Layer 0's input projections run on the sum(lengths) packed positions, the recurrent
            # products on the padded Tw steps (a row tile skips only the steps past its longest sentence); without attention only
            # encoder layer 0 is evaluated
            H, K, Tw, S1, att_b = 200, 512, int(ids.shape[1]), int(codes.shape[1]) - 1, att == "True"
            Hin = 2 * H if att_b else H
            mean_len = float(lengths.double().mean())
            enc = mean_len * (2 * 300 * 3 * H * 2) + Tw * (2 * H * 3 * H * 2) + \
                (Tw * (2 * 2 * H * 3 * H * 2 + 2 * H * 3 * H * 2 + 2 * H * H) if att_b else 0)
            dec = S1 * (2 * Hin * H + 2 * (2 * H * 3 * H * 2) + 2 * H * K + ((2 * H * H + 4 * Tw * H) if att_b else 0))
            fl = 3.0 * (enc + dec) * B
            tf = fl / (dt / n) / 1e12
            res["runs"].append({"att": att_b, "B": B, "ms_per_step": round(dt / n * 1e3, 4), "samples_per_s": round(B * n / dt, 1),
                                "loss": round(g.read_loss(), 4), "lost_replays": g.lost_replays, "lengths": "fixed across replays (baked into the graph)",
                                "flops_executed": fl, "achieved_TFLOPs": round(tf, 2),
                                "frac_of_f32_mfma_peak": round(tf / PEAK_F32_MFMA_TFLOPS, 4),
                                "decoder_steps": ("persistent cluster launches (code_cluster_fwd_kernel / code_cluster_bptt_kernel)" if served[1]
                                                  else "fused per-step kernels (g2v_attn_code_rollout_fwd / _bwd)" if served[0]
                                                  else "column-split per-operator kernels")})
            del g, net, opt
            import gc
            gc.collect()          # (a captured graph is gone before the next one is captured: ops.reset_side_streams)
    if with_cpu and keep is not None:
        try:
            from oracle import g2v_oracle as O
            sd, ids, lengths, codes = keep
            B, S, H = ids.shape[0], codes.shape[1], 200
            gm = torch.Generator().manual_seed(3)
            masks = {"emb": (torch.rand(S - 1, B, H, generator=gm) < 0.5).to(torch.uint8),
                     "dec_l0": (torch.rand(S - 1, B, H, generator=gm) < 0.8).to(torch.uint8)}
            cfg = dict(n_layers=2, dropout_prob=0.2, n_pre_poses=1, lr=5e-4, att=False)
            torch.set_num_threads(min(os.cpu_count() or 1, 8))
            adam = {}
            O.t2e_train_step(sd, adam, ids, lengths.long(), codes.long(), masks, cfg)
            ts = []
            while len(ts) < 10 and (sum(ts) < 10.0 or len(ts) < 3):
                t0 = time.perf_counter()
                O.t2e_train_step(sd, adam, ids, lengths.long(), codes.long(), masks, cfg)
                ts.append(time.perf_counter() - t0)
            res["cpu_baseline"] = {"value": round(B / statistics.median(ts), 1), "unit": "samples/s", "cores": min(os.cpu_count() or 1, 8),
                                   "kind": "port", "sample": f"median of {len(ts)} oracle t2e_train_step calls at B={B}, no attention"}
        except Exception as e:
            res["cpu_baseline"] = {"value": None, "error": f"{type(e).__name__}: {e}"[:200]}
    return res


def calibrate(lib):
    """What THIS box sustains (events on the launch stream): fp32 MFMA issue rate with no memory traffic, and a 2 x 1 GiB
    streaming copy (larger than the 256 MB Infinity Cache).  Reported next to the datasheet peaks, not instead of them."""
    from gesture2vec_amd._lib import check
    st = torch.cuda.current_stream()
    dev = torch.device("cuda", torch.cuda.current_device())
    blocks, iters = 256 * 8, 2000                    # 8 workgroups per CU = 8 waves per SIMD
    scratch = torch.empty(blocks * 256, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    check(lib.g2v_probe_mfma_f32(scratch.data_ptr(), blocks, 50, st.cuda_stream))
    e0.record(st)
    check(lib.g2v_probe_mfma_f32(scratch.data_ptr(), blocks, iters, st.cuda_stream))
    e1.record(st); e1.synchronize()
    mfma_tf = blocks * 4 * iters * 8 * 2048.0 / (e0.elapsed_time(e1) * 1e-3) / 1e12
    n = 1 << 28                                      # 1 GiB of floats
    src, dst = torch.empty(n, device=dev), torch.empty(n, device=dev)
    src.fill_(1.0)
    check(lib.g2v_probe_copy(src.data_ptr(), dst.data_ptr(), n, st.cuda_stream))
    e0.record(st)
    for _ in range(3):
        check(lib.g2v_probe_copy(src.data_ptr(), dst.data_ptr(), n, st.cuda_stream))
    e1.record(st); e1.synchronize()
    copy_gbs = 3 * 2 * 4.0 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del src, dst
    return {"mfma_f32_TFLOPs": round(mfma_tf, 1), "hbm_copy_GBps": round(copy_gbs, 0)}


def _cpu_baseline_worker(threads: int, B: int, n: int):
    """child of cpu_baseline: n oracle steps at B with `threads` threads, step times as one JSON line"""
    from oracle import g2v_oracle as O
    torch.set_num_threads(threads)
    T, D, H, K = CFG["T"], CFG["D"], CFG["H"], CFG["K"]
    p = CFG["dropout_prob"]
    cfg = dict(n_layers=2, dropout_prob=p, commitment_cost=CFG["beta"], n_pre_poses=1, conditioned=True,
               w_l1=CFG["w_l1"], w_cont=CFG["w_cont"], w_var=CFG["w_var"], lr=CFG["lr"])
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(B, T, D, generator=g)
    masks = {"dec": (torch.rand(T - 1, B, D, generator=g) < 0.05).to(torch.uint8)}
    if p > 0:
        masks["in"] = (torch.rand(T, B, D, generator=g) < 1 - p).to(torch.uint8)
        masks["enc_l0"] = (torch.rand(T, B, 2 * H, generator=g) < 1 - p).to(torch.uint8)
        masks["dec_l0"] = (torch.rand(T - 1, B, H, generator=g) < 1 - p).to(torch.uint8)
    sd, adam = O.init_vqvae_state(D, H, 2, K, seed=0), {}
    ts = []
    for i in range(n + 1):
        t0 = time.perf_counter()
        O.vqvae_train_step(sd, adam, x, masks, cfg)
        if i:
            ts.append(time.perf_counter() - t0)
    print(json.dumps(ts), flush=True)


def _train_iter_worker(B: int, n: int):
    """`--train-iter-worker B n`: the drop-in iteration itself -- train_iter_Autoencoder_VQ_seq2seq(args, epoch, x, x, net, optim)
    (reference train_eval/train_seq2seq.py:664-758), its per-iteration read-back of the loss included (the reference's
    loss.item()) -- at the contract shape, n timed iterations behind 20 untimed ones.  The contract line above times the same step
    as device-synchronised graph replays WITHOUT that read-back; this leg is what a user of the reference's training loop gets."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from gesture2vec_amd.model.Autoencoder_VQVAE_model import Autoencoder_VQVAE
    from gesture2vec_amd.train_eval.train_seq2seq import FusedClipAdam, train_iter_Autoencoder_VQ_seq2seq
    CFG.update({k: v for k, v in CONFIGS["full"].items() if k != "name"})
    torch.manual_seed(0)
    args = model_args()
    args.loss_l1_weight, args.loss_cont_weight, args.loss_var_weight, args.learning_rate = CFG["w_l1"], CFG["w_cont"], CFG["w_var"], CFG["lr"]
    net = Autoencoder_VQVAE(args, CFG["D"], CFG["T"]).to("cuda:0")
    net.train(True)
    optim = FusedClipAdam(net, CFG["lr"], betas=(0.5, 0.999))
    x = torch.randn(B, CFG["T"], CFG["D"], generator=torch.Generator().manual_seed(1234)).to("cuda:0")
    for _ in range(20):
        train_iter_Autoencoder_VQ_seq2seq(args, 1, x, x, net, optim)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        loss, perp = train_iter_Autoencoder_VQ_seq2seq(args, 1, x, x, net, optim)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"ms_per_iter": round(dt / n * 1e3, 4), "value": round(B * n / dt, 1), "unit": "chunks/s", "iters": n, "B": B,
                      "loss": round(float(loss["loss"]), 6),
                      "what": "train_iter_Autoencoder_VQ_seq2seq called in a Python loop, one 16-byte read-back per iteration "
                              "(loss, loss_vq, perplexity, fault latch) as the reference's loss.item()"}))


def main():
    if len(sys.argv) >= 4 and sys.argv[1] == "--train-iter-worker":
        return _train_iter_worker(int(sys.argv[2]), int(sys.argv[3]))
    if len(sys.argv) >= 5 and sys.argv[1] == "--cpu-baseline-worker":
        if len(sys.argv) >= 6:
            CFG.update({k: v for k, v in CONFIGS[sys.argv[5]].items() if k != "name"})
        return _cpu_baseline_worker(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 100 timed steps behind 20 untimed ones (0.2 s of GPU time; 30 / 5 read 0.5 % above the 300-step `sustained` figure on
    # every box of round 5 -- the first replays behind a short warm-up run at a lower clock)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="full", help="workload shape (see CONFIGS); the contract line is `full`")
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (weak scaling); default: the config's")
    ap.add_argument("--no-part-d", action="store_true", help="skip the text2embedding (Part d) samples/s object of the line")
    ap.add_argument("--no-graph", action="store_true", help="do not replay the step from a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sustained", type=int, default=300,
                    help="extra steps timed BEHIND the contract's --steps and reported as `sustained` (0 = off)")
    ap.add_argument("--dropout", type=float, default=None,
                    help="encoder-input / GRU inter-layer dropout_prob (config/VQ-VAE.yml ships 0.2; SURVEY.md 8(d) config 2 "
                         "and the default here use 0: the always-on Dropout(0.95) of the decoder input is drawn either way)")
    ap.add_argument("--wgrad-bf16x3", action="store_true",
                    help="opt-in: weight-gradient products as 3-term bf16 splits (reported in config.wgrad); default exact fp32")
    ap.add_argument("--no-loss-chase", action="store_true",
                    help="A/B: custom_loss as its own launch between the rollouts instead of the chaser kernel beside the forward rollout")
    ap.add_argument("--force-dp", action="store_true",
                    help="run the data-parallel code path (split graphs + RCCL all-reduce) even with one rank (diagnostic)")
    a = ap.parse_args()
    CFG.update({k: v for k, v in CONFIGS[a.config].items() if k != "name"})
    CFG["config"] = a.config
    if a.batch is None:
        a.batch = CFG["B"]
    if a.dropout is None:
        a.dropout = CFG["dropout_prob"]

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} needs torch.distributed.run with --nproc-per-node {a.gpus} (WORLD_SIZE={world})")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    use_dp = world > 1 or a.force_dp
    if use_dp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # "nccl" is RCCL on ROCm

    from gesture2vec_amd import _lib
    from gesture2vec_amd.model.Autoencoder_VQVAE_model import Autoencoder_VQVAE
    lib = _lib.load()                              # raises if the HIP library is not built
    assert lib.g2v_device_ok() == 1, "bench.py needs an MI355X (gfx950)"

    B, T, D = a.batch, CFG["T"], CFG["D"]
    CFG["dropout_prob"] = float(a.dropout)
    torch.manual_seed(0)                            # identical initial weights on every rank
    net = Autoencoder_VQVAE(model_args(), D, T).to(dev)
    net.rng_seed = 1234 + rank
    net.train(True)
    eng = net.engine()
    eng.wgrad_bf16x3 = bool(a.wgrad_bf16x3)
    eng.loss_chase = not a.no_loss_chase
    x = torch.randn(B, T, D, generator=torch.Generator().manual_seed(1234 + rank)).to(dev)   # rank-dependent shard

    reduce_fn = None
    if use_dp:
        from gesture2vec_amd.dp import GradStatsAllReduce, broadcast_state
        vq = net.vq_layer
        broadcast_state([eng.flat, vq._embedding.weight.data, vq._ema_w.data, vq._ema_cluster_size,
                         vq.pre_linear.weight.data, vq.pre_linear.bias.data])
        reduce_fn = GradStatsAllReduce()        # ONE RCCL all-reduce of [grads | EMA stats] per step

    kw = dict(w_l1=CFG["w_l1"], w_cont=CFG["w_cont"], w_var=CFG["w_var"], epoch=1, draw_masks=True)

    def local():
        eng.train_step_local(x, x, dp=use_dp, **kw)

    def apply():
        eng.train_step_apply(B, lr=CFG["lr"], world=world, dp=use_dp)

    def step():
        local()
        if use_dp:
            reduce_fn(eng.comm)
        apply()

    def barrier():
        if use_dp:
            dist.barrier()
        torch.cuda.synchronize()

    def capture(fn):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        return g

    # untimed warm-up (also sizes every workspace so that addresses are final before capture)
    for _ in range(max(a.warmup, 2)):
        step()
    torch.cuda.synchronize()
    graph = None
    dp_launch = "eager launches"
    run = step
    if not a.no_graph:
        try:
            if use_dp:
                # ONE hipGraph with the RCCL all-reduce captured inside it (one replay per step, nothing of the exchange on the
                # host); where the collective cannot be captured: two hipGraphs around it, [masks..backward] -> all-reduce(comm)
                # -> [EMA + clip/Adam]
                try:
                    g_step = capture(step)
                    run = g_step.replay
                    graph = g_step
                    dp_launch = "one hipGraph replay, RCCL all-reduce captured inside"
                except Exception as e:
                    print(f"[bench] all-reduce not capturable ({type(e).__name__}: {e}); two graphs around it", file=sys.stderr)
                    torch.cuda.synchronize()
                    g_local, g_apply = capture(local), capture(apply)

                    def run():
                        g_local.replay()
                        reduce_fn(eng.comm)
                        g_apply.replay()
                    graph = (g_local, g_apply)
                    dp_launch = "2 hipGraph replays around one RCCL all-reduce"
            else:
                g_step = capture(step)
                run = g_step.replay
                graph = g_step
            for _ in range(2):
                run()
            torch.cuda.synchronize()
        except Exception as e:   # capture is an optimisation, never a correctness requirement
            print(f"[bench] hipGraph capture unavailable ({type(e).__name__}: {e}); timing eager launches", file=sys.stderr)
            graph, run = None, step

    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        run()
    barrier()
    dt = time.perf_counter() - t0
    per_rank_ms = None
    if use_dp:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        allt = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(allt, tt)
        per_rank_ms = [round(float(v.item()) / a.steps * 1e3, 4) for v in allt]
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    # sustained rate: 300 more steps of the same replay, behind the contract's K steps (round-4 verdict: 20 x 1.6 ms is a 32 ms sample)
    sustained = None
    if a.sustained > 0:
        barrier()
        t1 = time.perf_counter()
        for _ in range(a.sustained):
            run()
        barrier()
        ds = time.perf_counter() - t1
        if use_dp:
            ts_ = torch.tensor([ds], device=dev, dtype=torch.float64)
            dist.all_reduce(ts_, op=dist.ReduceOp.MAX)
            ds = float(ts_.item())
        sustained = {"steps": a.sustained, "ms_per_step": round(ds / a.sustained * 1e3, 4),
                     "value": round(B * world * a.sustained / ds, 1), "unit": "chunks/s"}
    # data parallel: where a step's time goes per rank -- eager launches with events around the collective (the timed region
    # replays it from inside the graph): the local half, the all-reduce as exposed on the stream, the apply half
    dp_diag = None
    if use_dp:
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        acc = [0.0, 0.0, 0.0]
        n_diag = 10
        for _ in range(n_diag):
            dist.barrier()
            ev[0].record(); local(); ev[1].record(); reduce_fn(eng.comm); ev[2].record(); apply(); ev[3].record()
            torch.cuda.synchronize()
            for k in range(3):
                acc[k] += ev[k].elapsed_time(ev[k + 1])
        mine = torch.tensor([v / n_diag for v in acc], device=dev, dtype=torch.float64)
        alld = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(alld, mine)
        rows = [[round(float(v), 4) for v in r.tolist()] for r in alld]
        dp_diag = {"eager_steps": n_diag, "per_rank_ms_local_allreduce_apply": rows,
                   "allreduce_us_max": round(max(r[1] for r in rows) * 1e3, 1),
                   "local_skew_ms": round(max(r[0] for r in rows) - min(r[0] for r in rows), 4),
                   "timed_region_per_rank_ms_per_step": per_rank_ms,
                   "note": "the all-reduce interval includes waiting for the slowest rank's local half (skew) + RCCL's own latency"}
    comm_sums = None
    if use_dp:
        cs = torch.tensor([float(eng.comm.double().sum())], device=dev, dtype=torch.float64)
        allc = [torch.zeros_like(cs) for _ in range(world)]
        dist.all_gather(allc, cs)
        comm_sums = [float(c.item()) for c in allc]
    loss = eng.loss_terms[0].item() + eng.vq_scalars[0].item() / 400
    assert lib.g2v_dec_rollout_persist_fault(0) == 0, "the persistent rollout kernel latched a residency fault: the timed steps are invalid"
    assert loss == loss, "loss is NaN"

    out = None
    if rank == 0:
        out = {
            "metric": f"gesture-chunks/sec VQ-VAE fwd+bwd (T={CFG['T']},D={CFG['D']},K={CFG['K']})", "value": round(B * world * a.steps / dt, 1),
            "unit": "chunks/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{CONFIGS[a.config]['name']}: train_iter_Autoencoder_VQ_seq2seq on "
                                   f"synthetic N(0,1) pose chunks, B={B}/GPU, T={CFG['T']}, D={CFG['D']}, H={CFG['H']}, L=2 (E={2 * CFG['H']}), "
                                   f"K={CFG['K']}, dropout_prob={CFG['dropout_prob']:g} (+ always-on Dropout(0.95)), Adam lr={CFG['lr']:g}, "
                                   "random-init weights",
                       "name": a.config,
                       "global_batch": B * world, "per_gpu_batch": B,
                       "parallelism": f"dp{world}" if world > 1 else "single",
                       "launch": ("eager launches" if graph is None else dp_launch if use_dp else "hipGraph replay"),
                       "wgrad": "bf16x3 split products, f32 accumulate" if a.wgrad_bf16x3 else "f32",
                       "graph_branches_mask": int(eng.overlap),
                       "decoder_rollout": (lambda r: ("cluster: one persistent launch each way, (hidden-unit tile x row group) workgroups"
                                                      if lib.g2v_dec_rollout_cluster_ok(B, CFG["D"], CFG["H"]) else "one launch per time step")
                                           if r == 0 else
                                           f"persistent: one launch each way, {r} row tile(s) of 16 per workgroup")(
                                               int(lib.g2v_dec_rollout_tiles_per_workgroup(B, CFG["D"], CFG["H"]))),
                       "custom_loss": ("chaser kernel co-resident with the forward rollout + the backward rollout's tile load"
                                       if eng.buffers(B).get("loss_folded") else "own launch between the rollouts"),
                       "wgrad_inside_recurrent_kernels": {"decoder_mask_ih0_hh0_ih1_hh1": int(eng.buffers(B).get("fused_wgrad", 0)),
                                                          "encoder_mode": int(eng.buffers(B).get("enc_fused_wgrad", 0))},
                       "final_loss": round(loss, 6),
                       # what the collective really ran over: the process group's size, and the reduced comm buffer's checksum of
                       # the last step as every rank saw it (identical values = every rank applied the same update)
                       "dist_world_size": (dist.get_world_size() if use_dp else 1),
                       "reduced_comm_checksum_per_rank": comm_sums,
                       "dp_diag": dp_diag},
        }
        if sustained is not None:
            out["sustained"] = sustained
        if world == 1:
            out["roofline"] = vq_kernel_roofline(eng, B)
            # the whole step as EXECUTED (the dead encoder layer 1 of the reference is skipped): forward
            # flop per chunk x 3 (backward = data + weight gradients), against the same fp32 MFMA peak
            T, D, H, E, K = CFG["T"], CFG["D"], CFG["H"], eng.E, eng.K
            fl = 3.0 * (2 * T * D * H + 24 * T * H * H + (T - 1) * (4 * D * H + 24 * H * H) + 2 * E * E + 2 * K * E) * B
            tf = fl / (dt / a.steps) / 1e12
            out["roofline"]["whole_step"] = {"flops_executed": fl, "achieved": round(tf, 2), "unit": "TFLOP/s",
                                             "frac": round(tf / PEAK_F32_MFMA_TFLOPS, 4)}
            try:
                cal = calibrate(lib)
                out["roofline"]["measured_on_this_box"] = cal
                out["roofline"]["frac_of_measured_mfma"] = round(out["roofline"]["achieved"] / cal["mfma_f32_TFLOPs"], 4)
            except Exception as e:   # the calibration is context, never a reason to lose the bench line
                print(f"[bench] calibration probe failed ({type(e).__name__}: {e})", file=sys.stderr)
            if not a.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(B)
            if not a.no_part_d and a.config == "full":
                try:
                    out["text2embedding"] = part_d(with_cpu=not a.no_cpu_baseline)
                except Exception as e:   # an extra object of the line, never a reason to lose it
                    print(f"[bench] Part d failed ({type(e).__name__}: {e})", file=sys.stderr)
                    out["text2embedding"] = {"error": f"{type(e).__name__}: {e}"[:300]}
            if not a.no_part_d and a.config == "full":
                try:
                    out["bulk_assign"] = bulk_assign_line()
                except Exception as e:
                    print(f"[bench] bulk-assign leg failed ({type(e).__name__}: {e})", file=sys.stderr)
                    out["bulk_assign"] = {"error": f"{type(e).__name__}: {e}"[:300]}
            if not a.no_part_d and a.config == "full":
                try:
                    out["gru_resident"] = gru_resident_line()
                except Exception as e:
                    print(f"[bench] gru-resident leg failed ({type(e).__name__}: {e})", file=sys.stderr)
                    out["gru_resident"] = {"error": f"{type(e).__name__}: {e}"[:300]}
            if not a.no_part_d and a.config == "full" and world == 1 and not a.force_dp:
                # the configuration the reference SHIPS (config/VQ-VAE.yml: B = 128, T = 20, D = 40, H = 200), same step, as a child
                # process of this one (its own model, its own graph): an extra object of the line like Part d
                try:
                    import subprocess
                    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--config", "native", "--steps", "200", "--warmup", "10",
                                        "--no-cpu-baseline", "--no-part-d", "--sustained", "0"], capture_output=True, text=True, timeout=240)
                    d = json.loads(r.stdout.strip().splitlines()[-1])
                    out["shipped_config"] = {"workload": d["config"]["workload"], "ms_per_step": d["ms_per_step"], "value": d["value"],
                                             "unit": d["unit"], "steps": d["steps"], "launch": d["config"].get("launch")}
                except Exception as e:
                    print(f"[bench] shipped-config run failed ({type(e).__name__}: {e})", file=sys.stderr)
                    out["shipped_config"] = {"error": f"{type(e).__name__}: {e}"[:300]}
                # the drop-in iteration WITH its per-iteration read-back (round-5 verdict: the trainer gap was last measured in round 3)
                try:
                    import subprocess
                    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--train-iter-worker", str(B), "300"], capture_output=True,
                                       text=True, timeout=240)
                    d = json.loads(r.stdout.strip().splitlines()[-1])
                    d["gap_vs_graph_replay"] = round(d["ms_per_iter"] / (dt / a.steps * 1e3) - 1.0, 4)
                    out["train_iter"] = d
                except Exception as e:
                    print(f"[bench] train_iter leg failed ({type(e).__name__}: {e})", file=sys.stderr)
                    out["train_iter"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    if use_dp:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        # RCCL prints its version banner through C stdio, which is flushed at exit: flush it now so that the JSON line
        # is the LAST line on stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()

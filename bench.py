#!/usr/bin/env python3
"""bench.py -- rel_pose hot path on MI355X: image-pairs/sec, fwd+bwd, synthetic 384x384 pairs.

    python bench.py --gpus N --steps K --warmup W            (N>1: launched by torch.distributed.run, one rank per GPU)

Workload (BASELINE.json metric / configs[2] shape, "train.py ... batch 64, 1xMI355X fwd+bwd" on the synthetic
inputs of configs[1]): one train.py step = ViTEss.forward on [64,2,3,384,384] pairs per GPU (CNN front-end on
MIOpen, ViT + Essential Matrix Module + regressor on this repo's HIP kernels) + geodesic loss + backward +
(N>1: DDP gradient all-reduce over RCCL) + grad-clip + Adam step.  Inputs are resident in HBM before the timed
region.  One JSON line on rank 0; see DESIGN.md "Measurement" for how `roofline` and `cpu_baseline` are formed.
"""
import argparse
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # the host driver only supports dmabuf IPC: RCCL between ranks needs it
import rel_pose_amd._env  # noqa: F401,E402  (MIOpen user-db path; before torch / the first convolution)
import torch
import torch.distributed as dist
from rel_pose_amd import parallel

FP32_MFMA_PEAK_TFLOPS = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
BF16_MFMA_PEAK_TFLOPS = 2500.0       # same guide, dense bf16 MFMA (never the 2:1-sparsity figure)
HBM_PEAK_GBS = 8000.0                # same guide, HBM3E peak (spec); 6.29 TB/s is the measured copy rate
PRECISIONS = {"fp32": 0, "split3": 3, "bf16": 1}
FLOPS_FWD_PER_PAIR = 8588216320      # GEMM flops of the ViT+EMM+regressor hot path, forward
METRIC = "image-pairs/sec fwd+bwd @384x384, 1/2/4/8 MI355X; R,t err vs ref"
GRAPH_BELOW_PAIRS = 16               # single-GPU training steps at or below this many pairs replay captured HIP graphs by default


def model_args():
    return types.SimpleNamespace(noess="", pool_size=60, fc_hidden_size=512, fusion_transformer=True,
                                 transformer_depth=6, cross_features=False, use_single_softmax=False,
                                 no_pos_encoding=False, l1_pos_encoding=False)


def synthetic_batch(B, hw, device, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    images = torch.floor(torch.rand(B, 2, 3, hw, hw, generator=g) * 255.0)
    q = torch.randn(B, 4, generator=g)
    q = q / q.norm(dim=-1, keepdim=True)
    q = q * torch.where(q[:, 3:] < 0, -1.0, 1.0)
    t = torch.rand(B, 3, generator=g) * 2 - 1
    poses = torch.zeros(B, 2, 7)
    poses[:, :, 6] = 1.0
    poses[:, 1] = torch.cat([t, q], dim=-1)
    intr = torch.tensor([hw / 2.0] * 4).repeat(B, 2, 1)          # (fx,fy,cx,cy) = (192,...) at 384
    return images.to(device), poses.to(device), intr.to(device)


def cpu_baseline(hw, budget_s=20.0):
    """The oracle (CPU restatement, kind 'port') timed on this box's host cores: fwd+bwd of the same step
    (CNN + ViT/EMM incl. the reference's host positional-encoding loop + regressor + loss) on a bounded sample."""
    from oracle import relpose_oracle as O
    from rel_pose_amd.se3 import SE3
    from rel_pose_amd.losses import geodesic_loss_tensors
    ncpu = os.cpu_count() or 1
    Bc = 4
    shapes = dict(O.vit_param_shapes())
    shapes.update(O.cnn_param_shapes())
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v)
          for k, v in O.make_state(shapes).items()}
    images, poses, intr = synthetic_batch(Bc, hw, "cpu", 1)

    t_start = time.perf_counter()

    def step():
        i2 = intr.clone()
        x = O.preprocess(images)
        O.update_intrinsics(images.shape[-2:], i2)
        tokens = O.tokens_from_cnn(O.cnn_features(sd, x, train=True))
        pos = O.positional_encodings_loop(Bc, i2)                       # the reference's 576-iteration host loop
        xx = tokens + sd["fusion_transformer.pos_embed"]
        for l in range(5):
            xx = O.block(sd, "fusion_transformer.blocks.%d." % l, xx)
        xx = O.cross_block(sd, "fusion_transformer.blocks.5.", xx, i2, pos=pos)
        feats = O.layernorm(xx, sd["fusion_transformer.norm.weight"], sd["fusion_transformer.norm.bias"])
        Gs = SE3.IdentityLike(SE3(poses))
        est = O.normalize_preds(Gs.data, O.regress(sd, feats, Bc))
        ltr, lrot = geodesic_loss_tensors(SE3(poses), [SE3(est)])
        (10 * ltr + 10 * lrot).backward()
        for v in sd.values():
            if v.is_floating_point() and v.grad is not None:
                v.grad = None

    # pick the thread count that is best FOR THE CPU (oversubscribing a 256-thread host makes the oracle's many
    # small ops crawl): one timed step per candidate, then spend the budget on the winner
    best = None
    for th in sorted({min(ncpu, c) for c in (8, 16, 32, 64)}):
        torch.set_num_threads(th)
        step()
        t0 = time.perf_counter()
        step()
        dt = time.perf_counter() - t0
        if best is None or dt < best[1]:
            best = (th, dt)
        if time.perf_counter() - t_start > 0.5 * budget_s:
            break
    cores = best[0]
    torch.set_num_threads(cores)
    n, t0 = 0, time.perf_counter()
    while True:
        step()
        n += 1
        el = time.perf_counter() - t0
        if el > 0.5 * budget_s or n >= 50:
            break
    return {"value": round(Bc * n / el, 3), "unit": "image-pairs/sec", "cores": cores, "kind": "port",
            "sample": "%d steps of %d synthetic %dx%d pairs, fwd+bwd (CNN + ViT/EMM incl. host pos-enc loop + loss), "
                      "torch-CPU oracle, %d of %d host threads (best of 8/16/32/64), %.1f s"
                      % (n, Bc, hw, hw, cores, ncpu, el)}

TRAFFIC_FILES = ("r6_traffic.json", "r5_traffic.json", "r4_traffic.json", "r3_traffic.json", "r2_traffic.json")


def csrc_fingerprint():
    """sha256 (first 16 hex digits) over rel_pose_amd/csrc/*.hip and *.h -- tools/pmc_bench.sh stores the same value in the traffic
    file it writes, so a PMC figure measured on OTHER kernel sources is flagged instead of silently going stale."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "rel_pose_amd", "csrc")
    for fn in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h"))):
        with open(fn, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def pmc_traffic(kname, files=TRAFFIC_FILES):
    """(HBM bytes per launch of `kname`, source note, stale) from the committed rocprofv3 --pmc passes (they cannot run inside the timed
    process).  stale = the kernel sources have changed since the passes ran (or the file predates the fingerprint)."""
    for fn in files:
        tpath = os.path.join(ROOT, "profiles", fn)
        if os.path.exists(tpath):
            with open(tpath) as f:
                doc = json.load(f)
            ent = doc["kernels"].get(kname)
            if ent:
                stale = doc.get("csrc_sha16") != csrc_fingerprint()
                return (round(ent["hbm_bytes_per_launch"]), "profiles/%s (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE, separate passes)" % fn, stale)
    return None, None, None


def roofline_of(flops_per_launch, bytes_per_launch, seconds_per_launch, nl):
    """Which roofline bounds a kernel, from its ALGORITHMIC intensity against the machine balance of the pipe it runs on (matrix-pipe peak
    / HBM peak: 19.7 flop/B exact fp32, 312 flop/B bf16), and the achieved fraction of that roofline; the other pipe's figure rides along."""
    peak_tf = {0: FP32_MFMA_PEAK_TFLOPS, 3: BF16_MFMA_PEAK_TFLOPS / 6.0, 1: BF16_MFMA_PEAK_TFLOPS}[nl]
    t = max(seconds_per_launch, 1e-12)
    tf, gbs = flops_per_launch / t / 1e12, bytes_per_launch / t / 1e9
    mfma_bound = flops_per_launch / max(bytes_per_launch, 1.0) >= peak_tf * 1e12 / (HBM_PEAK_GBS * 1e9)
    if mfma_bound:
        return {"bound": "mfma", "achieved": round(tf, 2), "peak": round(peak_tf, 1), "unit": "TFLOP/s", "frac": round(tf / peak_tf, 4),
                "hbm_gbs_algorithmic": round(gbs, 1)}
    return {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
            "mfma_frac": round(tf / peak_tf, 4), "mfma_achieved_tflops": round(tf, 1)}


def supplementary_point(dev, tag, batch, hw, mode, precision, steps, warmup, traffic_files):
    """One bounded extra operating point in the SAME process, after the headline: its own model, its own
    roofline object (HIP-event-timed dominant kernel of that mode, found by survey_kernels), barrier-free (single GPU).  Never
    touches the headline fields."""
    from rel_pose_amd import ops
    from rel_pose_amd.losses import geodesic_loss_tensors
    from rel_pose_amd.model import ViTEss
    from rel_pose_amd.se3 import SE3
    nl = PRECISIONS[precision]
    saved = (ops.GEMM_PRECISION, ops.ATTN_BF16, ops.CNN_PRECISION, ops.TIMER)
    try:
        ops.set_gemm_precision(nl)
        ops.set_attention_precision(1 if nl == 1 else 0)
        ops.set_cnn_precision(1 if nl == 1 else 0)
        torch.manual_seed(0)
        model = ViTEss(model_args()).to(dev)
        for p in list(model.resnet.layer3.parameters()) + list(model.resnet.layer4.parameters()):
            p.requires_grad = False
        train = mode == "train"
        model.train(train)
        opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=5e-4, weight_decay=1e-5, fused=True) if train else None
        images, poses, intr = synthetic_batch(batch, hw, dev, 4321)
        Ps = SE3(poses)
        Gs = SE3.IdentityLike(Ps)

        def step():
            if train:
                opt.zero_grad(set_to_none=True)
                est = model(images, Gs, intrinsics=intr.clone())
                ltr, lrot = geodesic_loss_tensors(Ps, est)
                loss = 10.0 * ltr + 10.0 * lrot
                loss.backward()
                parallel.clip_grad_norm_(model.parameters(), 2.5)
                opt.step()
                return loss
            with torch.no_grad():
                return model(images, Gs, intrinsics=intr.clone())[0].data

        ops.TIMER = None
        step()
        torch.cuda.synchronize()
        dominant, survey = survey_kernels(step, 2, nl)
        kernel_symbol = tag_symbol(dominant, nl)
        timer = ops.KernelTimer(dominant)
        ops.TIMER = timer
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        timer.enabled = True
        t0 = time.perf_counter()
        last = None
        for _ in range(steps):
            last = step()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        timer.enabled = False
        n_launch, t_launch, flops = timer.summary()
        pairs = batch * steps
        tf = flops / max(n_launch, 1) / max(t_launch, 1e-12) / 1e12
        gbs = timer.bytes / max(n_launch, 1) / max(t_launch, 1e-12) / 1e9
        traffic, src, stale = pmc_traffic(kernel_symbol, traffic_files)
        roof = roofline_of(flops / max(n_launch, 1), timer.bytes / max(n_launch, 1), t_launch, nl)
        roof["survey"] = survey
        roof.update({"traffic": traffic, "traffic_unit": "HBM bytes per launch (PMC)", "traffic_source": src, "traffic_stale": stale,
                     "algorithmic_bytes_per_launch_avg": timer.bytes / max(n_launch, 1), "kernel": kernel_symbol,
                     "launches_timed": n_launch, "avg_launch_us": round(t_launch * 1e6, 2), "flops_per_launch_avg": flops / max(n_launch, 1),
                     "hot_path_tflops_whole_step": round((3 if train else 1) * FLOPS_FWD_PER_PAIR * pairs / el / 1e12, 2)})
        rec = {"value": round(pairs / el, 2), "unit": "image-pairs/sec", "ms_per_step": round(1e3 * el / steps, 3), "steps": steps,
               "warmup": warmup, "dtype": "bf16" if nl == 1 else "f32",
               "config": {"workload": tag, "pairs_per_gpu": batch, "mode": mode, "precision": precision,
                          "finite": bool(torch.isfinite(last).all())},
               "roofline": roof}
        del model, opt, images
        torch.cuda.empty_cache()
        return rec
    finally:
        ops.GEMM_PRECISION, ops.ATTN_BF16, ops.CNN_PRECISION, ops.TIMER = saved


def parse_instance(text):
    """'1,1,1,3' -> (1,1,1,3) (an rp_gemm instance); anything else is a kernel tag of rel_pose_amd.ops.timed"""
    try:
        return tuple(int(v) for v in text.split(","))
    except ValueError:
        return text


# ops.timed tag -> kernel symbol as rocprofv3 prints it at the bench's operating points (two-wave / three-wave workgroup forms of the
# full-size batches).  Only used to NAME the roofline kernel and to look its PMC traffic up; tests/test_bench_contract.py checks the
# name against the tracked kernel profile.
TAG_SYMBOLS = {
    "mlp_fused_fwd": "mlp_fused_kernel<4, 3, 0, false, true, false>", "mlp_fused_fwd_eval": "mlp_fused_kernel<4, 3, 0, false, false, false>",
    "mlp_fused_fwd_bf16": "mlp_fused_kernel<12, 3, 0, true, true, false>", "mlp_fused_fwd_eval_bf16": "mlp_fused_kernel<12, 3, 0, true, false, false>",
    "mlp_fused_bwd": "mlp_fused_kernel<12, 3, 1, false, false, false>", "mlp_fused_bwd_ln": "mlp_fused_kernel<12, 3, 1, false, false, true>",
    "mlp_fused_bwd_bf16": "mlp_fused_kernel<12, 3, 1, true, false, false>", "mlp_fused_bwd_ln_bf16": "mlp_fused_kernel<12, 3, 1, true, false, true>",
    "attn_fwd": "attn_fwd_kernel<2, false, 2, false, false, false>", "attn_fwd_savep": "attn_fwd_kernel<2, false, 2, false, false, true>",
    "attn_stats": "attn_fwd_kernel<3, true, 2, false, false, false>", "emm_stats": "attn_fwd_kernel<3, true, 2, false, true, false>",
    "attn_bwd_dkdv_p": "attn_bwd_dkdv_p_kernel<2, 2>", "attn_bwd_dkdv_ds": "attn_bwd_dkdv_kernel<2, 2, false>",
    "ds_matmul": "ds_matmul_kernel<3>", "ds_matmul_t": "ds_matmul_t_kernel<3>",
    "emm_apply": "emm_apply_kernel<false>", "emm_grad_ds": "emm_grad_kernel<false, false>",
    "emm_apply_s": "emm_apply_s_kernel", "emm_grad_ds_s": "emm_grad_kernel<false, true>",
    "linear_rows_ln": "linear_rows_kernel<true, false>", "linear_rows": "linear_rows_kernel<false, false>",
    "dw192_bf16": "dw192_bf16_kernel<false>", "dw192_bf16_f32b": "dw192_bf16_kernel<true>", "dw192_f32": "dw192_f32_kernel", "dw192_split3": "dw192_split3_kernel",
    "attn_fwd_bf16": "attn_fwd_bf16_kernel<2, false, 1>", "attn_stats_bf16": "attn_fwd_bf16_kernel<3, true, 1>", "attn_bwd_bf16": "attn_bwd_dkdv_bf16_kernel + attn_bwd_dq_bf16_kernel",
    "dx_lnbwd_bf16": "dx_lnbwd_bf16_kernel",
    "emm_apply_bf16": "emm_apply_bf16_kernel", "emm_grad_bf16": "emm_grad_bf16_kernel",
    "conv3x3_c64_f32": "conv3x3_c64_f32_kernel<false, false>", "conv3x3_c128_f32": "conv3x3_c128_f32_kernel", "conv_stem_fwd": "conv_stem_fwd_kernel", "conv3x3_c64_wgrad_f32": "conv3x3_c64_wgrad_f32_kernel", "conv_stem_wgrad_f32": "conv_stem_wgrad_f32_kernel",
    "conv3x3_c64_bf16": "conv3x3_c64_kernel", "conv3x3_c64_wgrad_bf16": "conv3x3_c64_wgrad_kernel", "conv_stem_fwd_bf16": "conv_stem_bf16_kernel",
    "conv_stem_wgrad_bf16": "conv_stem_wgrad_kernel",
}


# tags whose ONE C-ABI call launches several main kernels (the HIP events bracket the call): ranked by time per kernel, so that the
# `roofline` kernel stays what a per-kernel profile of the same step ranks first
KERNELS_PER_CALL = {"attn_bwd_bf16": 2}


def tag_symbol(tag, nl=0):
    """kernel symbol of an ops.timed tag or an rp_gemm instance tuple (exact-fp32 launches with whole 32-wide k-tiles run the
    LDS-DMA-staged kernel of csrc/gemm_dma.hip, the bf16-limb precisions the register-staged one of csrc/gemm.hip)"""
    if isinstance(tag, tuple):
        args = ", ".join(str(v) for v in tag)
        return ("gemm_dma_kernel<%s>" % args if nl == 0 and not os.environ.get("RP_GEMM_NO_DMA")
                else "gemm_kernel<%s, %d>" % (args, nl))
    return TAG_SYMBOLS.get(tag, tag)


def survey_kernels(step_fn, nsteps, nl):
    """The step's own MFMA kernels ranked by the time the step spends in them: `nsteps` UNTIMED steps with every tagged launch (and every
    rp_gemm instance) bracketed by HIP events on its launch stream (ops.KernelTimer, instance "*").  Returns (dominant tag, record):
    the record lists the top five and `hot_path_frac` = sum of algorithmic flops of ALL own MFMA kernels / sum of their time / the
    matrix-pipe peak -- the figure that tracks the whole hot path, where `roofline` describes its single largest kernel."""
    from rel_pose_amd import ops
    keep = ops.TIMER
    tm = ops.KernelTimer(ops.SURVEY)
    ops.TIMER = tm
    tm.enabled = True
    try:
        for _ in range(nsteps):
            step_fn()
        torch.cuda.synchronize()
    finally:
        tm.enabled = False
        ops.TIMER = keep
    rows = sorted(tm.survey(), key=lambda r: -r[2] / KERNELS_PER_CALL.get(r[0], 1))
    if not rows:
        return None, None
    peak = {0: FP32_MFMA_PEAK_TFLOPS, 3: BF16_MFMA_PEAK_TFLOPS / 6.0, 1: BF16_MFMA_PEAK_TFLOPS}[nl]
    tot_t, tot_f, tot_b = sum(r[2] for r in rows), sum(r[3] for r in rows), sum(r[4] for r in rows)
    top = [{"tag": (r[0] if isinstance(r[0], str) else "gemm" + str(list(r[0]))), "kernel": tag_symbol(r[0], nl), "launches_per_step": round(r[1] / nsteps, 2),
            "ms_per_step": round(1e3 * r[2] / nsteps, 4), "tflops": round(r[3] / max(r[2], 1e-12) / 1e12, 2),
            "mfma_frac": round(r[3] / max(r[2], 1e-12) / 1e12 / peak, 4),
            "hbm_gbs_algorithmic": round(r[4] / max(r[2], 1e-12) / 1e9, 1), "kernels_per_call": KERNELS_PER_CALL.get(r[0], 1)} for r in rows[:5]]
    rec = {"steps": nsteps, "own_mfma_kernels_ms_per_step": round(1e3 * tot_t / nsteps, 3), "top": top,
           "hot_path_frac": round(tot_f / max(tot_t, 1e-12) / 1e12 / peak, 4),
           "hot_path_tflops": round(tot_f / max(tot_t, 1e-12) / 1e12, 2),
           "hot_path_hbm_gbs_algorithmic": round(tot_b / max(tot_t, 1e-12) / 1e9, 1),
           "note": "HIP-event time of every own MFMA kernel over %d untimed steps before the judged region; `roofline` times the first "
                   "entry over the judged steps" % nsteps}
    return rows[0][0], rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="image pairs per GPU")
    ap.add_argument("--hw", type=int, default=384)
    ap.add_argument("--mode", default="train", choices=("train", "fwd"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-supplementary", action="store_true",
                    help="skip the two extra bounded points (BASELINE configs[1] forward-only at 64 pairs, configs[4] bf16 at 128 pairs "
                         "per GPU) that the default single-GPU run times after the headline loop and attaches as `supplementary`")
    ap.add_argument("--graph", action="store_true", default=None,
                    help="replay the step from captured HIP graphs (rel_pose_amd/graph.py) instead of launching eagerly.  Default: graphs "
                         "for single-GPU training at <= %d pairs per GPU -- there the ~350 launches of a step are host-bound when issued "
                         "eagerly (6 pairs: 740-900 pairs/s from run to run, 892 +- 1 replayed) -- and eager + DDP above, where replay "
                         "measured neutral" % GRAPH_BELOW_PAIRS)
    ap.add_argument("--no-graph", dest="graph", action="store_false", help="always launch eagerly")
    ap.add_argument("--scope", default="full", choices=("full", "hot"),
                    help="full = images -> CNN -> hot path (the metric); hot = synthetic CNN maps -> hot path only "
                         "(kernel profiling; not the headline number)")
    ap.add_argument("--timer-instance", default=None,
                    help="kernel timed for `roofline`: default = the own MFMA kernel the step spends most time in, found by a survey of two "
                         "untimed steps (every tagged launch HIP-event-timed); or force a gemm_kernel<aL,bL,TM,TN> instance (\"1,1,1,3\") / a "
                         "kernel tag of ops.timed (\"dw192_f32\")")
    ap.add_argument("--precision", default="fp32", choices=tuple(PRECISIONS),
                    help="how rp_gemm multiplies its fp32 operands: fp32 = exact v_mfma_f32_32x32x2_f32 (default); split3 = "
                         "three bf16 limbs per operand, six limb products on the bf16 MFMA pipe, fp32-grade results; "
                         "bf16 = operands rounded to bf16 (BASELINE.json configs[4]; NOT the headline metric)")
    args = ap.parse_args()
    if args.graph is None:
        args.graph = (args.mode == "train" and args.batch <= GRAPH_BELOW_PAIRS and int(os.environ.get("WORLD_SIZE", "1")) == 1
                      and args.scope == "full" and not os.environ.get("RP_BENCH_FORCE_DIST"))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node N" % (args.gpus, world))
    if os.environ.get("RP_BENCH_SHARE_GPU"):      # functional test of the N>1 path on a 1-GPU box (gloo, all ranks on cuda:0)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # RP_BENCH_FORCE_DIST: run the N>1 code path (process group on RCCL, DDP wrapper, bucketed gradient all-reduce, barrier +
    # max-over-ranks timing) with a single rank -- the only way to put RCCL under this code on a 1-GPU box
    force_dist = bool(os.environ.get("RP_BENCH_FORCE_DIST"))
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group(backend=os.environ.get("RP_DIST_BACKEND", "nccl"), init_method="env://",
                                world_size=world, rank=rank)

    # CNN front-end: the convolution solvers for the bench shapes (128 images of 224x224 per GPU) come from the shipped MIOpen
    # user db (rel_pose_amd/_env.py), immediate mode, no search.  Other --batch / --hw values are not in it:
    # RP_CUDNN_BENCHMARK=1 lets MIOpen search them once, inside the untimed priming step below (takes tens of seconds).
    torch.backends.cudnn.benchmark = os.environ.get("RP_CUDNN_BENCHMARK", "0") == "1"
    from rel_pose_amd import _lib, ops
    from rel_pose_amd.losses import geodesic_loss_tensors
    from rel_pose_amd.model import ViTEss
    from rel_pose_amd.se3 import SE3
    _lib.load()

    torch.manual_seed(0)
    model = ViTEss(model_args()).to(dev)
    for p in list(model.resnet.layer3.parameters()) + list(model.resnet.layer4.parameters()):
        p.requires_grad = False                                        # reference train.py:60-64
    train = args.mode == "train"
    model.train(train)
    net = model
    if (world > 1 or force_dist) and train and not args.graph:
        net = parallel.wrap(model, device_ids=[local_rank])
        # (reference train.py:66-67; bucketed all-reduce of the 19.26 M trainable grads overlaps the backward)
    elif world > 1:
        for t in list(model.parameters()) + list(model.buffers()):      # what DDP's constructor does: rank 0's replica everywhere
            dist.broadcast(t.data, 0)
    graphed = train and args.graph
    if graphed:
        net = model                                                    # the graphs do their own single flat all-reduce
    # fused=True: the reference's torch.optim.Adam update (train.py:69) as one multi-tensor kernel instead of ~10
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=5e-4, weight_decay=1e-5,
                           capturable=graphed, fused=True)
    images, poses, intr = synthetic_batch(args.batch, args.hw, dev, 1234 + rank)
    Ps = SE3(poses)
    Gs = SE3.IdentityLike(Ps)

    hot = args.scope == "hot"
    if hot:
        if world > 1:
            raise SystemExit("--scope hot is a single-GPU profiling aid")
        fmap = torch.rand(2 * args.batch, 192, 24, 24, device=dev)      # post-ReLU-like CNN map
        i24 = (intr * (24.0 / args.hw)).contiguous()

        def net(_images, Gs_, intrinsics=None):                          # noqa: F811
            return [SE3(model.forward_tokens(fmap, Gs_.data, i24))]

    def step():
        if train:
            opt.zero_grad(set_to_none=True)
            est = net(images, Gs, intrinsics=intr.clone())
            ltr, lrot = geodesic_loss_tensors(Ps, est)
            loss = 10.0 * ltr + 10.0 * lrot
            loss.backward()
            parallel.clip_grad_norm_(model.parameters(), 2.5)
            opt.step()
            return loss
        with torch.no_grad():
            return net(images, Gs, intrinsics=intr.clone())[0].data

    ops.set_gemm_precision(PRECISIONS[args.precision])
    ops.set_attention_precision(1 if args.precision == "bf16" else 0)      # configs[4]: bf16 MFMA attention / EMM GEMMs too
    ops.set_cnn_precision(1 if args.precision == "bf16" and not os.environ.get("RP_BF16_KEEP_FP32_CNN") else 0)   # and the MIOpen convolutions
    nl = PRECISIONS[args.precision]
    timer = ops.KernelTimer(parse_instance(args.timer_instance) if args.timer_instance else "none")
    ops.TIMER = timer
    eager_step = step
    # untimed priming step, even with --warmup 0: lazy init and MIOpen's solver search never fall in the timed steps.  The kernel timer is ON
    # for it: the first timed hipEventRecord on a stream switches its HSA queue to profiling mode and creates the runtime's signal pool
    # (50-90 ms once, measured: 10 timed steps read 39-44 ms per step instead of 34.6 when that fell into the timed region)
    timer.enabled = not os.environ.get("RP_NO_TIMER")
    eager_step()
    torch.cuda.synchronize()
    timer.enabled = False
    timer.reset()
    survey = None
    if args.timer_instance is None:
        # which own kernel dominates THIS step on THIS box: two untimed steps, every tagged launch timed; `roofline` then follows it
        dominant, survey = survey_kernels(eager_step, 2, nl)
        if world > 1 or force_dist:      # every rank must time the same symbol: rank 0's choice
            box = [dominant]
            dist.broadcast_object_list(box, src=0)
            dominant = box[0]
        timer = ops.KernelTimer(dominant if dominant is not None else "none")
        ops.TIMER = timer
    if rank == 0 and not hot:
        rel_pose_amd._env.check_db()          # warns when the shipped MIOpen solver db does not belong to the loaded MIOpen
    if graphed:
        from rel_pose_amd.graph import GraphedTrainStep
        fwd = (lambda im, G, it: net(im, G, intrinsics=it))
        gs = GraphedTrainStep(model, opt, images, poses, intr, forward_fn=fwd).capture(warmup=max(args.warmup, 3))
        step = gs.step
    for _ in range(args.warmup):
        step()

    def fence():
        torch.cuda.synchronize()
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()

    roctx = None
    if os.environ.get("RP_ROCTX"):       # rocprofv3 --selected-regions: profile only the timed steps
        import ctypes
        roctx = ctypes.CDLL("librocprofiler-sdk-roctx.so")
    fence()
    if roctx is not None:
        roctx.roctxProfilerResume(0)
    timer.enabled = not os.environ.get("RP_NO_TIMER")
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = step()
    fence()
    el = time.perf_counter() - t0
    if roctx is not None:
        roctx.roctxProfilerPause(0)
    timer.enabled = False
    dist_diag = None
    if world > 1 or force_dist:
        # self-diagnosing scaling record: every rank's own time for the timed steps, and -- AFTER the timed
        # region, never inside it -- the same step without the gradient exchange (DDP no_sync), so that the all-reduce time the
        # backward does not hide can be read off the line: exposed = ms_per_step - ms_per_step_no_allreduce (max over ranks each)
        el, per_rank = parallel.gather_step_times(el, args.steps, dev)
        dist_diag = {"per_rank_ms_per_step": per_rank}
        if train and not graphed and hasattr(net, "no_sync"):
            k = max(3, min(10, args.steps))
            # the probe runs whole steps (optimizer included) WITHOUT the gradient exchange: the replicas diverge.  Parameters, buffers and
            # the optimizer state are put back afterwards, so anything that runs later in this process sees consistent replicas.
            keep_model = {n: t.detach().clone() for n, t in model.state_dict().items()}
            keep_opt = {i: {kk: (vv.detach().clone() if torch.is_tensor(vv) else vv) for kk, vv in st.items()}
                        for i, st in enumerate(opt.state.values())}
            fence()
            t1 = time.perf_counter()
            with net.no_sync():
                for _ in range(k):
                    step()
            fence()
            ns = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
            dist.all_reduce(ns, op=dist.ReduceOp.MAX)
            with torch.no_grad():
                for n, t in model.state_dict().items():
                    t.copy_(keep_model[n])
                for i, st in enumerate(opt.state.values()):
                    for kk, vv in st.items():
                        if torch.is_tensor(vv):
                            vv.copy_(keep_opt[i][kk])
            opt.zero_grad(set_to_none=True)
            ops.invalidate_pad_cache()
            ms_ns = 1e3 * float(ns.item()) / k
            dist_diag.update({"ms_per_step_no_allreduce": round(ms_ns, 3), "steps_no_allreduce": k,
                              "exposed_allreduce_ms": round(1e3 * el / args.steps - ms_ns, 3),
                              "gradient_bytes": int(sum(p.numel() for p in model.parameters() if p.requires_grad) * 4),
                              "note": "no_allreduce = the same step under DDP.no_sync(), timed after the judged region; parameters, "
                                      "buffers and optimizer state restored afterwards"})
    finite = bool(torch.isfinite(last).all())

    if graphed:
        # per-kernel HIP events cannot be recorded inside a graph replay: time the roofline kernel over eager steps of the
        # SAME step function, same resident batch, right after the timed region (rocprofv3 over this command sees both)
        fence()
        timer.enabled = True
        for _ in range(max(2, min(5, args.steps))):
            gs._fwd_bwd()
            gs._exchange()
            gs._update()
        fence()
        timer.enabled = False
    if rank == 0:
        n_launch, t_launch, flops = timer.summary()
        achieved = flops / max(n_launch, 1) / max(t_launch, 1e-12) / 1e12
        pairs = world * args.batch * args.steps
        traffic, traffic_src, traffic_stale = None, None, None
        kname = tag_symbol(timer.instance, nl)
        # matrix-pipe ceiling of the timed kernel in ALGORITHMIC (2MNK) flops: the exact-fp32 MFMA peak, or the dense
        # bf16 MFMA peak divided by the limb products issued per fp32 product (6 for split3, 1 for bf16)
        peak = {0: FP32_MFMA_PEAK_TFLOPS, 3: BF16_MFMA_PEAK_TFLOPS / 6.0, 1: BF16_MFMA_PEAK_TFLOPS}[nl]
        peak_note = {0: "fp32 MFMA peak (v_mfma_f32_32x32x2_f32)",
                     3: "dense bf16 MFMA peak 2500 TF / 6 limb products per fp32 multiply-add (v_mfma_f32_32x32x16_bf16); "
                        "the same kernel is %.2f of the 157.3 TF fp32-MFMA peak it replaces" % (achieved / FP32_MFMA_PEAK_TFLOPS),
                     1: "dense bf16 MFMA peak"}[nl]
        # PMC passes cannot run inside the timed process (tools/pmc_bench.sh); the committed passes of the matching operating point
        # (64 pairs fp32 training / forward, 128 pairs bf16 training; bytes per launch of any other batch were not measured: null)
        tfiles = None
        if args.hw == 384:
            if args.batch == 64 and nl == 0:
                tfiles = TRAFFIC_FILES if train else ("r6_traffic_fwd.json", "r5_traffic_fwd.json", "r4_traffic_fwd.json", "r3_traffic_fwd.json")
            elif args.batch == 128 and nl == 1 and train:
                tfiles = ("r6_traffic_bf16.json", "r5_traffic_bf16.json", "r4_traffic_bf16.json", "r3_traffic_bf16.json")
        traffic, traffic_src, traffic_stale = pmc_traffic(kname, tfiles) if tfiles else (None, None, None)
        rec = {
            "metric": METRIC, "value": round(pairs / el, 2), "unit": "image-pairs/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * el / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if nl == 1 else "f32",
            "data": "synthetic",
            "config": {"workload": ("train.py step (ViTEss fwd + geodesic loss + bwd + grad all-reduce + clip + Adam)"
                                    if train else "ViTEss.forward, eval, no_grad") +
                                   ", synthetic %dx%d pairs" % (args.hw, args.hw),
                       "scope": args.scope, "pairs_per_gpu": args.batch, "global_batch_pairs": world * args.batch,
                       "parallelism": "dp%d" % world, "finite": finite,
                       "gemm_operand_precision": {0: "exact fp32 MFMA", 3: "fp32 operands split into 3 bf16 limbs, 6 limb "
                                                  "products on the bf16 MFMA pipe, fp32 accumulate (fp32-grade: measured "
                                                  "error vs fp64 <= the fp32-MFMA kernel's)", 1: "bf16 operands, fp32 accumulate: "
                                                  "Linear GEMMs (rp_gemm precision 1), the attention / EMM contractions "
                                                  "(v_mfma_f32_32x32x16_bf16) and the CNN front-end's MIOpen convolutions"}[nl],
                       "launch": "HIP graph replay (fwd+loss+bwd | flat grad all-reduce | clip+Adam)" if graphed else "eager",
                       "hot_path_share": "ViT+EMM+regressor on HIP kernels; ResNet front-end on MIOpen (SURVEY 8f-1)"},
            "roofline": dict(roofline_of(flops / max(n_launch, 1), timer.bytes / max(n_launch, 1), t_launch, nl),
                             peak_note=peak_note, traffic=traffic, traffic_unit="HBM bytes per launch (PMC)",
                             traffic_source=traffic_src, traffic_stale=traffic_stale,
                             algorithmic_bytes_per_launch_avg=timer.bytes / max(n_launch, 1), kernel=kname,
                             kernel_chosen_by=("survey: largest total HIP-event time among the step's own MFMA kernels" if survey is not None
                                               else "--timer-instance"),
                             launches_timed=n_launch, avg_launch_us=round(t_launch * 1e6, 2),
                             flops_per_launch_avg=flops / max(n_launch, 1),
                             hot_path_frac=(survey or {}).get("hot_path_frac"),
                             hot_path_tflops_whole_step=round((3 if train else 1) * FLOPS_FWD_PER_PAIR * pairs / el / 1e12, 2),
                             survey=survey),
        }
        if (world == 1 and not force_dist and not args.no_supplementary and train and args.scope == "full" and args.batch == 64
                and args.precision == "fp32" and not graphed):
            # the default single-GPU run also times BASELINE configs[1] (forward only, 64 pairs) and configs[4]'s per-GPU workload
            # (bf16, 128 pairs) in this process, bounded to a few seconds each; the headline fields above are already final
            sup = {}         # (the headline model stays resident: 288 GB of HBM make freeing it pointless)
            for key, kw in (("fwd_only", dict(tag="ViTEss.forward, eval, no_grad, synthetic %dx%d pairs (BASELINE configs[1])" % (args.hw, args.hw),
                                              batch=64, mode="fwd", precision="fp32", steps=60, warmup=5,
                                              traffic_files=("r6_traffic_fwd.json", "r5_traffic_fwd.json", "r4_traffic_fwd.json", "r3_traffic_fwd.json"))),
                            ("bf16_128", dict(tag="train.py step, bf16 MFMA operands in Linear / attention / EMM GEMMs and the MIOpen "
                                                  "convolutions, fp32 accumulate (BASELINE configs[4] per-GPU workload)",
                                              batch=128, mode="train", precision="bf16", steps=20, warmup=3,
                                              traffic_files=("r6_traffic_bf16.json", "r5_traffic_bf16.json", "r4_traffic_bf16.json", "r3_traffic_bf16.json")))):
                try:
                    sup[key] = supplementary_point(dev, hw=args.hw, **kw)
                except Exception as e:          # a supplementary point must never take the headline line down with it
                    sup[key] = {"error": "%s: %s" % (type(e).__name__, e)}
            rec["supplementary"] = sup
        if dist_diag is not None:
            rec["distributed"] = dist_diag
        if not args.no_cpu_baseline and world == 1:
            rec["cpu_baseline"] = cpu_baseline(args.hw)
        print(json.dumps(rec), flush=True)
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

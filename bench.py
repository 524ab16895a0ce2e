#!/usr/bin/env python3
"""Headline benchmark: user-sequences/sec (fwd+bwd+Adam) of the CAST training hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N=1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N>1)

Workload (BASELINE.json configs[1]): ml-1m-shaped synthetic corpus, CAST1 (models/cast_1.py), maxlen 200,
hidden 50, 2 blocks, 1 head, dropout 0.2, batch 128 per GPU (weak scaling: global batch 128*N is one
batch of the single reference sampler stream, rank r takes rows [128r, 128(r+1))), random-init weights.
A step = forward + backward + (RCCL all-reduce for N>1) + dense TF-Adam, replayed from a
HIP graph; the id batches are resident in HBM before the timed region.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, HIP-event timed) and `cpu_baseline`
(the oracle's torch-CPU restatement of the same step on the host cores; rank 0, N=1 only)."""
import argparse
import json
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import castrec_amd  # noqa: E402
from castrec_amd import engine as E  # noqa: E402
from castrec_amd import lib as L  # noqa: E402
from castrec_amd import synth  # noqa: E402
from castrec_amd.sampler import WarpSampler  # noqa: E402

PEAK = {"hbm": (8000.0, "GB/s"), "mfma": (157.3, "TFLOP/s")}     # MI355X_MICROARCH.md: HBM3E 8 TB/s; fp32-input MFMA 157.3 TF


def hyper(args):
    return E.Hyper(maxlen=args.maxlen, hidden_units=args.hidden_units, num_blocks=args.num_blocks, num_heads=args.num_heads,
                   dropout_rate=args.dropout_rate, max_bins=200, lr=1e-3, seed=42)


def algo_work(name, fnargs, eng):
    """(flops, bytes) one launch does ALGORITHMICALLY -- None where not modelled."""
    import ctypes as C
    B, T, D, H, M = eng.B, eng.T, eng.D, eng.H, eng.M
    if name == "cr_attn_fwd":
        return 2.0 * D * T * (T + 1) * B, None, "mfma"            # QK^T + PV, causal half, all heads
    if name == "cr_attn_bwd":
        return 2.0 * 2.0 * D * T * (T + 1) * B, None, "mfma"      # backward = 2x forward
    if name == "cr_gemm_rows":
        arr, n = fnargs[0], fnargs[1]
        return sum(2.0 * arr[i].M * arr[i].N * arr[i].K for i in range(n)), None, "mfma"
    if name == "cr_gemm_wgrad":
        arr, n = fnargs[0], fnargs[1]
        return sum(2.0 * arr[i].M * arr[i].N * arr[i].K for i in range(n)), None, "mfma"
    if name in ("cr_layernorm_fwd", "cr_eltwise"):
        return None, 2.0 * M * D * 4, "hbm"
    if name == "cr_layernorm_bwd":
        return None, 3.0 * M * D * 4, "hbm"
    if name == "cr_embed_fwd":
        return None, (2.0 * M * D * 4 + M * 4), "hbm"             # table rows read + activation write + ids
    if name == "cr_embed_bwd":
        return None, (3.0 * M * D * 4 + M * 4), "hbm"             # grad read + row read-modify-write
    if name == "cr_head_fwd_bwd":
        return None, (2.0 * M * D * 4 + M * 8) + 2.0 * M * D * 4 + 2.0 * 2 * M * D * 4, "hbm"
    if name == "cr_adam_step":
        return None, 7.0 * 4 * eng.layout.n_total, "hbm"
    return None, None, "hbm"


def kernel_profile(eng, staged, n_steps=8):
    """Eager steps with a HIP event pair around every launch (events on the launch stream).  Each step starts
    with a ~0.5 ms device-side sleep so the host runs ahead of the GPU: every launch is already queued when the
    GPU reaches its start event, and an event pair brackets kernel execution, not host launch latency (without
    it the pairs read ~5 us long against rocprofv3's kernel durations)."""
    evs = []
    stream = torch.cuda.current_stream()
    s = stream.cuda_stream
    prog = eng.fwd + eng.bwd + [eng._adam]
    eng.Gflat.zero_()
    for it in range(n_steps + 2):
        eng.ids_all.copy_(staged[it % staged.shape[0]])
        torch.cuda._sleep(1_000_000)
        rec = []
        for name, fn, a in prog:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            rc = fn(*a, s)
            e1.record(stream)
            if rc != 0:
                raise RuntimeError(name)
            rec.append((name, a, e0, e1))
        torch.cuda.synchronize()
        if it >= 2:
            evs.append([(n, a, e0.elapsed_time(e1) * 1e3) for n, a, e0, e1 in rec])
    agg = {}
    for step in evs:
        for idx, (n, a, us) in enumerate(step):
            k = (idx, n)
            agg.setdefault(k, []).append(us)
    per_launch = []
    for (idx, n), v in sorted(agg.items()):
        a = prog[idx][2]
        fl, by, bound = algo_work(n, a, eng)
        per_launch.append(dict(idx=idx, name=n, us=float(np.mean(v)), flops=fl, bytes=by, bound=bound))
    by_name = {}
    for r in per_launch:
        d = by_name.setdefault(r["name"], dict(us=0.0, launches=0, flops=0.0, bytes=0.0, bound=r["bound"]))
        d["us"] += r["us"]; d["launches"] += 1
        d["flops"] += r["flops"] or 0.0; d["bytes"] += r["bytes"] or 0.0
    return per_launch, by_name


KERNELS_OF = {"cr_attn_fwd": ["k_attn_fwd"], "cr_attn_bwd": ["k_attn_bwd"],
              "cr_layernorm_fwd": ["k_ln_fwd"], "cr_layernorm_bwd": ["k_ln_bwd"], "cr_adam_step": ["k_adam"],
              "cr_head_fwd_bwd": ["k_head"], "cr_embed_fwd": ["k_embed_fwd"], "cr_embed_bwd": ["k_embed_bwd"]}


def pmc_traffic(abi_name):
    """HBM bytes per launch of the device kernels behind a C-ABI entry, from the committed rocprofv3 --pmc
    summary of this same command (profiles/*_pmc_summary.json, tools/pmc_summary.py); None if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")))
    if not files:
        return None, "n/a"
    with open(files[-1]) as f:
        summ = json.load(f)
    want = KERNELS_OF.get(abi_name, ["k_" + abi_name[3:]])
    tot, hit = 0, 0
    for k, e in summ.items():
        if any(k.startswith(w) for w in want):
            tot += e["hbm_traffic_bytes"]; hit += 1
    return (tot if hit else None), os.path.relpath(files[-1], ROOT)


def cpu_baseline(args, batches, usernum, itemnum, budget_s=20.0):
    """The oracle's torch-CPU restatement of the same training step (fp32, all host threads)."""
    from oracle import fpmodel as fm
    ohp = fm.Hyper(maxlen=args.maxlen, hidden_units=args.hidden_units, num_blocks=args.num_blocks, num_heads=args.num_heads,
                   dropout_rate=args.dropout_rate, max_bins=200, lr=1e-3)
    P = fm.init_params(args.model, usernum, itemnum, ohp, seed=0, dtype=torch.float32)
    opt = fm.AdamTF(P, lr=1e-3)
    B = args.batch_size
    g = torch.Generator().manual_seed(0)
    drop = lambda site, shape: (torch.rand(shape, generator=g) >= args.dropout_rate)
    n, t_total = 0, 0.0
    for i in range(1 + 50):
        b = batches[i % len(batches)]
        batch = fm.to_batch(*[x[:B] for x in b])
        t0 = time.time()
        out, G = fm.loss_and_grads(args.model, P, ohp, batch, drop)
        P = opt.step(P, G)
        dt = time.time() - t0
        if i >= 1:
            n += 1; t_total += dt
            if t_total > budget_s or n >= 20:
                break
    return dict(value=round(n * B / t_total, 1), unit="sequences/s", cores=torch.get_num_threads(), kind="port",
                sample="%d steps of the same B=%d T=%d %s step (fwd+bwd+Adam) in torch-CPU fp32 = oracle/fpmodel.py; "
                       "cpu_count=%d" % (n, B, args.maxlen, args.model, os.cpu_count()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--model", default="cast_1")
    ap.add_argument("--batch_size", type=int, default=128, help="per GPU")
    ap.add_argument("--maxlen", type=int, default=200)
    ap.add_argument("--hidden_units", type=int, default=50)
    ap.add_argument("--num_blocks", type=int, default=2)
    ap.add_argument("--num_heads", type=int, default=1)
    ap.add_argument("--dropout_rate", type=float, default=0.2)
    ap.add_argument("--n_slabs", type=int, default=0, help="gradient slabs; 0 = engine default (ceil(M/128) capped at 256)")
    ap.add_argument("--attn_precision", default=None, choices=["f32", "bf16x3", "bf16"],
                    help="arithmetic of the attention products (default: the engine's, bf16x3 = bf16 MFMA on hi+lo split operands)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-json", default=None, help="write the per-kernel HIP-event table here")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus))
    torch.cuda.set_device(local_rank)
    dist = None
    force_dist = os.environ.get("CASTREC_FORCE_DIST") == "1"      # exercise the RCCL path with a single rank
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    B, T = args.batch_size, args.maxlen
    Bg = B * world
    corpus = synth.preset("ml-1m")
    sargs = types.SimpleNamespace(seed=42, bin_in_hours=48, max_bins=200, log_scale=False)
    smp = WarpSampler(sargs, corpus, corpus.usernum, corpus.itemnum, batch_size=Bg, maxlen=T)
    NB = 16
    host_batches = []
    for _ in range(NB):
        u, seq, pos, neg, ts, rat, hrs, dys, _ = smp.next_batch()
        sl = slice(rank * B, (rank + 1) * B)
        host_batches.append((seq[sl], pos[sl], neg[sl], ts[sl], hrs[sl], dys[sl]))
    smp.close()
    staged = torch.from_numpy(np.stack([np.stack([a.reshape(-1) for a in hb]) for hb in host_batches]).astype(np.int32)).cuda()

    eng = E.Engine(args.model, corpus.usernum, corpus.itemnum, hyper(args), B, training=True, n_slabs=args.n_slabs,
                   batch_global=Bg, row_offset=rank * B * T, attn_precision=args.attn_precision)
    dp = dist is not None
    if dp:
        dist.broadcast(eng.P, 0)
    use_graph = not args.no_graph
    if use_graph:
        eng.ids_all.copy_(staged[0])
        eng.capture(dp=dp)
        eng.set_step(1); eng.Mom.zero_(); eng.Vel.zero_(); eng.Gflat.zero_()

    def step(i):
        eng.ids_all.copy_(staged[i % NB])
        if not dp:
            if use_graph:
                eng.graph.launch()
            else:
                eng.launch_step()
        else:
            if use_graph:
                eng.graph.launch()
            else:
                eng.launch_backward_to_flat()
            dist.all_reduce(eng.Gflat)                       # RCCL over xGMI: table + dense grads + loss stats, one bucket
            eng.launch_adam_from_flat()

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt[0])
    loss, auc = eng.loss_auc()

    out = None
    if rank == 0:
        per_launch, by_name = kernel_profile(eng, staged)
        dom = max(by_name.items(), key=lambda kv: kv[1]["us"])
        name, d = dom
        bound = d["bound"]
        peak, unit = PEAK[bound]
        if bound == "mfma":
            achieved = d["flops"] / (d["us"] * 1e-6) / 1e12
        else:
            achieved = d["bytes"] / (d["us"] * 1e-6) / 1e9
        traffic, src = pmc_traffic(name)
        roofline = dict(kernel=name, bound=bound, achieved=round(achieved, 3), peak=peak, unit=unit,
                        frac=round(achieved / peak, 5), traffic=traffic, launches_per_step=d["launches"],
                        us_per_launch=round(d["us"] / d["launches"], 2),
                        algorithmic_per_launch=(d["flops"] if bound == "mfma" else d["bytes"]) / d["launches"],
                        note="achieved = algorithmic %s per launch / average HIP-event duration of its %d launches per step "
                             "(eager instrumented pass, 8 steps); traffic = HBM bytes per launch, (2*FETCH_SIZE + WRITE_SIZE) KiB "
                             "from %s" % ("flops" if bound == "mfma" else "bytes", d["launches"], src))
        if args.profile_json:
            with open(args.profile_json, "w") as f:
                json.dump(dict(per_launch=per_launch, by_name=by_name, n_launches=eng.n_launches()), f, indent=1, default=str)
        out = {
            "metric": "user-sequences/sec (fwd+bwd)", "value": round(Bg * args.steps / dt, 1), "unit": "sequences/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "ml-1m-shaped synthetic (6040 users, 3416 items), %s maxlen=%d hidden_units=%d num_blocks=%d "
                                   "num_heads=%d dropout=%.2f, batch %d/GPU (global %d), fwd+bwd+dense TF-Adam per step"
                                   % (args.model, T, args.hidden_units, args.num_blocks, args.num_heads, args.dropout_rate, B, Bg),
                       "parallelism": "dp%d" % world, "hip_graph": use_graph, "launches_per_step": eng.n_launches(),
                       "final_loss": round(loss, 5), "final_auc": round(auc, 5)},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, host_batches, corpus.usernum, corpus.itemnum)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Headline benchmark: user-sequences/sec (fwd+bwd+Adam) of the CAST training hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
        N = 1: runs in this process.  N > 1 without WORLD_SIZE in the environment: this process touches no GPU, it
        spawns N fresh rank processes of itself (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set)
        and relays rank 0's JSON line.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (the driver's form)

Workload (BASELINE.json configs[1]): ml-1m-shaped synthetic corpus, CAST1 (models/cast_1.py), maxlen 200,
hidden 50, 2 blocks, 1 head, dropout 0.2, batch 128 per GPU (weak scaling: global batch 128*N is one
batch of the single reference sampler stream, rank r takes rows [128r, 128(r+1))), random-init weights.
A step = forward + backward + (RCCL all-reduce for N>1) + dense TF-Adam, replayed from a
HIP graph; the id batches are resident in HBM before the timed region.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, HIP-event timed) and `cpu_baseline`
(the oracle's torch-CPU restatement of the same step on the host cores; rank 0, N=1 only)."""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# MI355X_MICROARCH.md: HBM3E 8 TB/s; fp32-input MFMA 157.3 TF (the fp32-grade arithmetic of "f32" and of the bf16x3 split
# form is priced against it); dense bf16 MFMA ~2.5 PF (plain bf16)
PEAK = {"hbm": (8000.0, "GB/s"), "mfma": (157.3, "TFLOP/s")}
PEAK_BF16_TF = 2500.0


def launch_ranks(n, argv, env=None, timeout=None):
    """Starts n fresh rank processes `python <argv>` on this node (one per GPU, rendezvous on 127.0.0.1) and waits.
    The caller must not have touched the GPU; nothing is re-exec'd.  Rank 0 inherits stdout; returns the worst exit code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        e = dict(os.environ if env is None else env)
        e.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")           # dmabuf IPC (RCCL across processes on this driver)
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=e, stdout=None if r == 0 else subprocess.DEVNULL))
    # poll all ranks: when one dies (or the deadline passes) the rest -- blocked in a collective the dead rank will never
    # join -- are killed instead of being waited for until the process-group timeout
    rc = 0
    deadline = time.time() + (timeout if timeout is not None else 3600.0)
    live = list(procs)
    while live:
        for p in list(live):
            r = p.poll()
            if r is not None:
                live.remove(p)
                rc = max(rc, abs(r))
        if live and (rc != 0 or time.time() > deadline):
            for p in live:
                p.kill()
            for p in live:
                p.wait()
            rc = max(rc, 1)
            break
        if live:
            time.sleep(0.05)
    return rc


def launcher_selftest():
    """A rank of `bench.py --launcher-selftest --gpus N`: gloo all-reduce on the CPU, rank 0 prints one JSON line.
    tests/test_bench_launcher.py drives launch_ranks() through it without a GPU."""
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo")
    t = torch.tensor([float(dist.get_rank() + 1)])
    dist.all_reduce(t)
    if dist.get_rank() == 0:
        print(json.dumps({"selftest": float(t[0]), "world": dist.get_world_size(), "backend": dist.get_backend()}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


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
    if name == "cr_head_fwd_bwd_ln":
        # the training-path pos / neg gather: two table rows per batch row + its ids, the sequence-embedding row and the final
        # LayerNorm's input row read, the LayerNorm-backward row written; then either the two logit derivatives per row (occurrence
        # index: the table gradient is gathered in cr_adam_step) or the scatter's read-modify-write of two table-gradient rows
        scatter = 2.0 * 2 * M * D * 4 if fnargs[0]._obj.table_grad else 8.0 * M
        return None, 2.0 * M * D * 4 + 8.0 * M + 2.0 * M * D * 4 + 1.0 * M * D * 4 + scatter, "hbm"
    if name == "cr_adam_step":
        d = fnargs[0]._obj
        if d.tg:
            # occurrence index: p / m / v read and written for every parameter; the table gradient is never stored -- its rows are
            # gathered: per occurrence one gradient row (seq: two partials) + the occurrence word, per pos / neg one coefficient
            g = d.tg.contents
            rows = (2.0 if g.rows2 else 1.0) * M + 2.0 * M + (M if g.lay.T_pos else 0)
            return None, 6.0 * 4 * eng.layout.n_total + 4.0 * eng.layout.n_dense + rows * D * 4 + 4.0 * 4 * M, "hbm"
        return None, 7.0 * 4 * eng.layout.n_total, "hbm"
    # fused row-phase kernels: GEMM flops of the phases each one holds
    if name in ("cr_block_ln_qkv_fwd", "cr_block_ln_qkv_fwd_gather"):
        return 2.0 * M * D * 3 * D, None, "mfma"
    if name == "cr_block_ln_ffn_fwd":
        return 2.0 * M * D * 2 * D, None, "mfma"
    if name == "cr_block_ln_ffn_fwd_tail":
        return 2.0 * M * D * 2 * D + (2.0 * M * D * 3 * D if fnargs[1]._obj.kind == 1 else 0.0), None, "mfma"
    if name == "cr_block_ln_ffn_bwd":
        return 2.0 * M * D * 4 * D, None, "mfma"
    if name in ("cr_block_ln_qkv_bwd", "cr_block_ln_qkv_bwd_scatter", "cr_stack_qkv_bwd", "cr_stack_qkv_bwd_scatter"):
        return 2.0 * M * D * 6 * D, None, "mfma"
    if name in ("cr_stack_ffn_bwd", "cr_stack_ffn_bwd_ln", "cr_stack_ffn_bwd_heads"):
        return 2.0 * M * D * 4 * D, None, "mfma"
    if name == "cr_stack_block_bwd":
        # one launch per block: feed-forward backward (2 data + 2 weight gradients) + attention backward (2 x forward: SURVEY 8d)
        # + Q / K / V projections backward (3 data + 3 weight gradients).  ALGORITHMIC work: the chain both workgroups of a
        # sequence run (the feed-forward data gradients, twice) is counted once
        return 2.0 * M * D * 4 * D + 2.0 * 2.0 * D * T * (T + 1) * B + 2.0 * M * D * 6 * D, None, "mfma"
    # hidden sizes 128 / 192 / 256 (cr_wide.hip): the activations do not fit the caches at these sizes (M x D x 4 = 13 MB per
    # tensor at config C4, ~20 tensors per block), so the row phases are priced against HBM: rows read + rows written
    if name == "cr_wide_ln_qkv_fwd":
        return 2.0 * M * D * 3 * D, 5.0 * M * D * 4, "hbm"        # x in; q_in, Q, K, V out
    if name == "cr_wide_ln_ffn_fwd":
        return 2.0 * M * D * 2 * D, 4.0 * M * D * 4, "hbm"        # o in; f_in, hid, y out
    if name == "cr_wide_ln_ffn_bwd":
        own = bool(fnargs[0]._obj.g_w1)
        return 2.0 * M * D * (4 if own else 2) * D, (9.0 if own else 7.0) * M * D * 4, "hbm"   # dy, hid, o (+ f_in, g rows again) in; g2, g1, d_o out
    if name == "cr_wide_ln_qkv_bwd":
        own = bool(fnargs[0]._obj.g_wqkv)
        return 2.0 * M * D * (6 if own else 3) * D, (11.0 if own else 6.0) * M * D * 4, "hbm"  # dQ dK dV d_o x (+ q_in, x, dQ dK dV again) in; dx out
    if name in ("cr_stack_fwd", "cr_stack_fwd_head"):
        # (cr_stack_fwd_head: + the prediction head on the last launch's rows -- row gathers and dot products, no matrix work counted)
        # per block: Q K V projections + causal attention (QK^T + PV, causal half) + the two feed-forward layers
        nb = fnargs[0]._obj.n_blocks
        return nb * (2.0 * M * D * 3 * D + 2.0 * D * T * (T + 1) * B + 2.0 * M * D * 2 * D), None, "mfma"
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
        eng.load_slot(staged[eng.step_number() % staged.shape[0]])        # (slot = step number mod slots: where Adam looks for the batch's occurrence index)
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
        # (cr_stack_fwd_head = cr_stack_fwd with the prediction head riding on its last launch: one entry, the stack forward's)
        d = by_name.setdefault("cr_stack_fwd" if r["name"] == "cr_stack_fwd_head" else r["name"], dict(us=0.0, launches=0, flops=0.0, bytes=0.0, bound=r["bound"]))
        d["us"] += r["us"]; d["launches"] += 1
        d["flops"] += r["flops"] or 0.0; d["bytes"] += r["bytes"] or 0.0
    return per_launch, by_name


KERNELS_OF = {"cr_attn_fwd": ["k_attn_fwd", "k_bf_fwd"], "cr_attn_bwd": ["k_attn_bwd", "k_bf_bwd"],
              "cr_layernorm_fwd": ["k_ln_fwd"], "cr_layernorm_bwd": ["k_ln_bwd"], "cr_adam_step": ["k_adam"],
              "cr_head_fwd_bwd": ["k_head"], "cr_embed_fwd": ["k_embed_fwd"], "cr_embed_bwd": ["k_embed_bwd"],
              "cr_block_ln_qkv_fwd": ["k_block_ln_qkv_fwd"], "cr_block_ln_qkv_fwd_gather": ["k_block_ln_qkv_fwd"],
              "cr_block_ln_ffn_fwd": ["k_block_ln_ffn_fwd"], "cr_block_ln_ffn_fwd_tail": ["k_block_ln_ffn_fwd"],
              "cr_block_ln_ffn_bwd": ["k_block_ln_ffn_bwd"], "cr_block_ln_qkv_bwd": ["k_block_ln_qkv_bwd"],
              "cr_block_ln_qkv_bwd_scatter": ["k_block_ln_qkv_bwd"], "cr_stack_fwd": ["k_stack_fwd"], "cr_stack_fwd_head": ["k_stack_fwd"],
              "cr_stack_ffn_bwd": ["k_stack_ffn_bwd"], "cr_stack_ffn_bwd_ln": ["k_stack_ffn_bwd"], "cr_stack_ffn_bwd_heads": ["k_stack_ffn_bwd"], "cr_stack_qkv_bwd": ["k_stack_qkv_bwd"], "cr_stack_qkv_bwd_scatter": ["k_stack_qkv_bwd"], "cr_stack_block_bwd": ["k_stack_block_bwd"],
              "cr_wide_ln_qkv_fwd": ["k_wide_qkv_fwd"], "cr_wide_ln_ffn_fwd": ["k_wide_ffn_fwd"], "cr_wide_ln_ffn_bwd": ["k_wide_ffn_bwd"],
              "cr_wide_ln_qkv_bwd": ["k_wide_qkv_bwd"], "cr_gemm_rows": ["k_gemm_rows"], "cr_gemm_wgrad": ["k_gemm_wgrad"]}


HEADLINE = dict(model="cast_1", batch_size=128, maxlen=200, hidden_units=50, num_blocks=2, num_heads=1, corpus="ml-1m", lazy_adam=False)


def config_tag(args):
    """'' for the headline workload, else a tag naming the shape (profiles/r*_<tag>_<precision>_pmc_summary.json)."""
    if all(getattr(args, k) == v for k, v in HEADLINE.items()):
        return ""
    return "%s_T%d_D%d_L%d_H%d_B%d_%s%s" % (args.model, args.maxlen, args.hidden_units, args.num_blocks, args.num_heads, args.batch_size,
                                             args.corpus.replace("-", ""), "_lazy" if args.lazy_adam else "")


def pmc_lookup(abi_name, precision, tag=""):
    """Counter evidence for the device kernels behind a C-ABI entry.  NOT measured in this run: read from the committed
    rocprofv3 --pmc summary of this same command AND workload (profiles/r*[_<tag>]_<precision>_pmc_summary.json, written by
    tools/prof.sh + tools/pmc_summary.py; tag = config_tag(): empty for the headline workload): HBM bytes per launch
    ((2*FETCH_SIZE + WRITE_SIZE) KiB, MI355X_MICROARCH.md section HBM) and the MFMA-busy fraction
    (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)).  (None, None, reason) when no summary of this
    workload is committed -- a summary of another workload is never substituted."""
    import glob
    import re
    pat = re.compile(r"^r\d+[a-z]?_%s%s_pmc_summary\.json$" % ((re.escape(tag) + "_") if tag else "", re.escape(precision)))
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")) if pat.match(os.path.basename(f)))
    if not files:
        return None, None, "no committed PMC summary for this workload (%s)" % (tag or "headline")
    with open(files[-1]) as f:
        summ = json.load(f)
    want = KERNELS_OF.get(abi_name, ["k_" + abi_name[3:]])
    tot, busy, act, hit = 0, 0.0, 0.0, 0
    for k, e in summ.items():
        if any(k.startswith(w) for w in want):
            tot += e["hbm_traffic_bytes"]; hit += 1
            busy += e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0); act += e.get("GRBM_GUI_ACTIVE", 0.0)
    if not hit:
        return None, None, os.path.relpath(files[-1], ROOT) + " (kernel absent)"
    return tot, (round(busy / (act / 8.0 * 1024.0), 4) if act else None), os.path.relpath(files[-1], ROOT)


def executed_tile_fraction(host_batches, T):
    """Share of the ALGORITHMIC attention work (causal half, all T positions) that the kernels execute on these batches:
    they skip 16 x 16 (query, key) tiles that hold only padding keys or lie above the diagonal.  The contract's
    `frac` divides the algorithmic flops by the time, so it over-states how busy the matrix pipe is by 1 / this."""
    import numpy as np
    nkt = (T + 15) // 16
    live, full = 0.0, 0.0
    for hb in host_batches:
        seq = hb[0]
        for row in seq:
            nz = np.flatnonzero(row)
            kf = (nz[0] // 16) if len(nz) else nkt
            n = nkt - kf
            live += n * (n + 1) / 2.0 * 256.0
            full += T * (T + 1) / 2.0
    return round(live / full, 4)


def cpu_info():
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return model


def cpu_baseline(model, maxlen, args, batches, usernum, itemnum, budget_s=25.0, min_steps=20):
    """The oracle's torch-CPU restatement of the same training step (fp32, all host threads)."""
    import torch
    from oracle import fpmodel as fm
    ohp = fm.Hyper(maxlen=maxlen, hidden_units=args.hidden_units, num_blocks=args.num_blocks, num_heads=args.num_heads,
                   dropout_rate=args.dropout_rate, max_bins=200, lr=1e-3)
    P = fm.init_params(model, usernum, itemnum, ohp, seed=0, dtype=torch.float32)
    opt = fm.AdamTF(P, lr=1e-3)
    B = args.batch_size
    g = torch.Generator().manual_seed(0)
    drop = lambda site, shape: (torch.rand(shape, generator=g) >= args.dropout_rate)
    n, t_total = 0, 0.0
    for i in range(1 + 400):
        b = batches[i % len(batches)]
        batch = fm.to_batch(*[x[:B, -maxlen:] for x in b])
        t0 = time.time()
        out, G = fm.loss_and_grads(model, P, ohp, batch, drop)
        P = opt.step(P, G)
        dt = time.time() - t0
        if i >= 1:
            n += 1; t_total += dt
            if n >= min_steps and (t_total > budget_s or n >= 4 * min_steps):
                break
            if t_total > 3 * budget_s:
                break
    return dict(value=round(n * B / t_total, 1), unit="sequences/s", cores=torch.get_num_threads(), kind="port",
                sample="%d steps of the B=%d T=%d %s step (fwd+bwd+Adam) in torch-CPU fp32 = oracle/fpmodel.py; %d threads of %d "
                       "logical CPUs, %s" % (n, B, maxlen, model, torch.get_num_threads(), os.cpu_count(), cpu_info()))


OTHER_CONFIGS = [
    ("C1 configs[0]: ml-1m SASRec maxlen=50 hidden_units=50 num_blocks=2 num_heads=1", ["--model", "sasrec", "--maxlen", "50"]),
    ("C3 configs[2] shape: Beauty-sized vocabulary (57 289 items, Zipf 1.1), SASRec maxlen=50 hidden_units=64 num_heads=2",
     ["--model", "sasrec", "--maxlen", "50", "--hidden_units", "64", "--num_heads", "2", "--dropout_rate", "0.5", "--corpus", "beauty"]),
    ("C4 configs[3] shape on one GPU: SASRec maxlen=200 hidden_units=128 num_blocks=4 num_heads=4, ml-1m-sized table",
     ["--model", "sasrec", "--maxlen", "200", "--hidden_units", "128", "--num_heads", "4", "--num_blocks", "4"]),
    ("C5 configs[4] shape on one GPU at batch 32: SASRec maxlen=512 hidden_units=256 num_blocks=2 num_heads=4, ml-1m-sized table "
     "(the 10 M-item table: --corpus c5 [--lazy_adam], profiles/r02d_c5_*_bench.json)",
     ["--model", "sasrec", "--maxlen", "512", "--hidden_units", "256", "--num_heads", "4", "--num_blocks", "2", "--batch_size", "32"]),
    # the same two shapes at their full vocabularies: the tables no longer fit the caches and Adam sweeps them
    ("C4 configs[3] at full size on one GPU: Books-sized corpus (600 000 users, 368 000 items, Zipf 1.0), SASRec maxlen=200 hidden_units=128 "
     "num_blocks=4 num_heads=4, batch 128",
     ["--model", "sasrec", "--maxlen", "200", "--hidden_units", "128", "--num_heads", "4", "--num_blocks", "4", "--corpus", "books", "--steps", "60"]),
    ("C5 configs[4] at full size on one GPU: 10 M-item table (10.24 GB fp32 + two Adam moments + gradient = 41 GB resident), SASRec maxlen=512 "
     "hidden_units=256 num_blocks=2 num_heads=4, batch 128, dense TF-Adam (the reference's optimiser); corpus cut: 4 000 of the 10^6 users "
     "are materialised (the sampler draws users uniformly: a step's work does not depend on the user count)",
     ["--model", "sasrec", "--maxlen", "512", "--hidden_units", "256", "--num_heads", "4", "--num_blocks", "2", "--corpus", "c5", "--steps", "12", "--warmup", "3"]),
    ("C5 configs[4] at full size, row-sparse (lazy) Adam on the item table -- a DEVIATION from the reference's dense update (DESIGN.md "
     "section 8); same corpus cut",
     ["--model", "sasrec", "--maxlen", "512", "--hidden_units", "256", "--num_heads", "4", "--num_blocks", "2", "--corpus", "c5", "--lazy_adam",
      "--steps", "30", "--warmup", "5"]),
]


def other_configs(steps=100, warmup=10):
    """The other BASELINE.json shapes through this same program, one short child run each (not the headline line: sequences/s,
    ms per step and launches per step of fwd + bwd + dense TF-Adam at batch 128 per GPU unless the label says otherwise)."""
    import subprocess
    res = []
    for label, flags in OTHER_CONFIGS:
        cmd = [sys.executable, os.path.abspath(__file__), "--no-cpu-baseline", "--no-gather", "--no-extra-precisions", "--no-other-configs",
               "--steps", str(steps), "--warmup", str(warmup)] + flags              # (a config's own --steps / --warmup come last and win)
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
            d = json.loads(r.stdout.strip().splitlines()[-1])
            res.append(dict(config=label, value=d["value"], unit=d["unit"], ms_per_step=d["ms_per_step"],
                            launches_per_step=d["config"]["launches_per_step"], attn_precision=d["config"]["attn_precision"],
                            dominant=dict(kernel=d["roofline"]["kernel"], bound=d["roofline"]["bound"], frac=d["roofline"]["frac"])))
        except Exception as e:                                # a failed side run must not cost the headline line
            res.append(dict(config=label, error=repr(e)[:200]))
    return res


def gather_block(reps=24):
    """HBM roofline of the item-embedding gather at config C5's table (10 M x 256 fp32 = 10.24 GB, far beyond the 256 MiB
    Infinity Cache), uniformly random rows.  (a) cr_embed_fwd on one C5 step (65 536 rows: table row x sqrt(D) + positional
    row, masked, written as the fp32 activation -- as many bytes written as read); (b) the read-only form, cr_test_logits
    (rows gathered and reduced against the sequence embedding, nothing but 4 bytes per row written).

    EVERY launch gathers a fresh set of rows (its own id tensor): round 2 replayed one id tensor, so the 67 MB of table
    rows of launch k were still in the Infinity Cache for launch k + 1 and the HIP-event time (73 us) was a cache number;
    rocprofv3 of the interleaved kernels read 93-100 us.  With fresh rows both clocks agree."""
    import numpy as np
    import torch
    from castrec_amd import ops as O
    V, D, T = 10_000_000, 256, 512
    table = torch.empty(V, D, device="cuda", dtype=torch.float32).uniform_(-0.01, 0.01)
    rs = np.random.RandomState(0)
    NW = 3

    def timed(f):
        for k in range(NW):
            f(k)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(reps):
            f(NW + k)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps

    M = 128 * T
    ids = torch.from_numpy(rs.randint(1, V, (NW + reps, M)).astype(np.int32)).cuda()
    out = torch.empty(M, D, device="cuda")
    pos = torch.randn(T, D, device="cuda")
    us_a = timed(lambda k: O.embed_fwd(ids[k], table, T, out, D, scale=float(D) ** 0.5, pos_table=pos, mask_ids=ids[k]))
    Bq, nc = 4096, 101
    cand = torch.from_numpy(rs.randint(1, V, (NW + reps, Bq, nc)).astype(np.int32)).cuda()
    semb = torch.randn(Bq, D, device="cuda")
    logits = torch.empty(Bq, nc, device="cuda")
    us_b = timed(lambda k: O.test_logits(semb, D, table, cand[k], Bq, 1, D, logits))
    rd_a, wr_a = M * (D * 4 + 4), M * D * 4
    rd_b = Bq * nc * (D * 4 + 4)
    # (c) the head kernel of the training step (cr_head_fwd_bwd_ln: sasrec.py:87-101 + the final LayerNorm's backward) at the same
    # table: the pos / neg rows of one C5 step, fresh ids per launch; the table gradient goes through the occurrence index, so the
    # kernel's table traffic is these two reads per batch row
    import ctypes as C
    from castrec_amd import lib as L
    pn = torch.from_numpy(rs.randint(1, V, (NW + reps, 2, M)).astype(np.int32)).cuda()
    semb_h, x_h, dx_h = torch.randn(M, D, device="cuda"), torch.randn(M, D, device="cuda"), torch.empty(M, D, device="cuda")
    gam = torch.ones(D, device="cuda")
    n_sl = 256
    slabs = torch.zeros(n_sl, 2 * D, device="cuda")
    coef = torch.empty(2, M, device="cuda")
    hstate = torch.zeros(16, device="cuda")

    def head(k):
        hd = L.HeadDesc(semb_h.data_ptr(), D, table.data_ptr(), pn[k, 0].data_ptr(), pn[k, 1].data_ptr(), M, D, V, hstate.data_ptr(), None, 0,
                        None, None, None, coef.data_ptr())
        nd = L.LnBwdDesc(x_h.data_ptr(), D, gam.data_ptr(), None, 0, dx_h.data_ptr(), D, 0, slabs.data_ptr(), slabs.data_ptr() + 4 * D, 2 * D, n_sl, M, D, 1e-8)
        L.call("cr_head_fwd_bwd_ln", C.byref(hd), C.byref(nd), torch.cuda.current_stream().cuda_stream)
    us_c = timed(head)
    rd_c_table = 2 * M * (D * 4 + 4)
    all_c = rd_c_table + 2 * M * D * 4 + M * D * 4 + 8 * M
    res = dict(table="10M x 256 fp32 (10.24 GB), uniform rows, a fresh row set per launch", peak_GBps=8000.0, launches_timed=reps,
               embed_fwd=dict(rows=M, us=round(us_a, 1), read_GBps=round(rd_a / us_a / 1e3, 1), read_write_GBps=round((rd_a + wr_a) / us_a / 1e3, 1),
                              read_frac=round(rd_a / us_a / 1e3 / 8000.0, 4), read_write_frac=round((rd_a + wr_a) / us_a / 1e3 / 8000.0, 4)),
               read_only=dict(kernel="cr_test_logits", rows=Bq * nc, us=round(us_b, 1), read_GBps=round(rd_b / us_b / 1e3, 1),
                              read_frac=round(rd_b / us_b / 1e3 / 8000.0, 4)),
               head_ln=dict(kernel="cr_head_fwd_bwd_ln", table_rows=2 * M, us=round(us_c, 1), table_read_GBps=round(rd_c_table / us_c / 1e3, 1),
                            table_read_frac=round(rd_c_table / us_c / 1e3 / 8000.0, 4), all_bytes_GBps=round(all_c / us_c / 1e3, 1),
                            all_bytes_frac=round(all_c / us_c / 1e3 / 8000.0, 4),
                            note="the training step's pos / neg gather: per batch row two table rows (fresh: HBM) + the sequence-embedding row "
                                 "and the LayerNorm input row (streamed) read, one gradient row written; all_bytes_frac is the number to hold "
                                 "against the roof"),
               note="MI355X_MICROARCH.md: float4 copy ceiling 6.29 TB/s (79 % of the 8 TB/s spec); random whole-row gathers into "
                    "registers 5.5-5.8 TB/s.  embed_fwd (the training-path gather) writes as many bytes as it reads: read_write_frac is "
                    "its number to hold against the roof; the read-only form is the one to hold against the >= 70 % READ target")
    # the same two kernels under rocprofv3 (committed summary of tools/gather_pmc.py, a fresh row set per launch as here): profiled
    # passes clock lower (MI355X_MICROARCH.md, DVFS item 2), so these fractions are the conservative ones to quote
    prof = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_gather_pmc_summary.json")))
    if prof:
        try:
            pj = json.load(open(prof[-1]))
            for key, kern, nbytes in (("read_only", "k_test_logits", rd_b), ("embed_fwd", "k_embed_fwd", rd_a + wr_a), ("head_ln", "k_head_ln", all_c)):
                ent = [v for k, v in pj.items() if k.startswith(kern)]
                if ent:
                    us = ent[0]["avg_ns"] / 1e3
                    res[key]["rocprof"] = dict(source=os.path.basename(prof[-1]), us=round(us, 2), frac=round(nbytes / us / 1e3 / 8000.0, 4),
                                               hbm_traffic_bytes=ent[0].get("hbm_traffic_bytes"), algorithmic_bytes=nbytes)
        except Exception as e:
            res["rocprof_error"] = repr(e)[:120]
    del table, out
    torch.cuda.empty_cache()
    return res


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
    ap.add_argument("--no-gather", action="store_true", help="skip the C5-table gather block (allocates 10.24 GB)")
    ap.add_argument("--no-extra-precisions", action="store_true", help="skip the short runs in the other attention precisions")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short runs of the other BASELINE.json shapes (child processes)")
    ap.add_argument("--corpus", default="ml-1m", choices=["ml-1m", "beauty", "books", "c5"],
                    help="synthetic corpus preset (castrec_amd/synth.py); the headline line is ml-1m")
    ap.add_argument("--lazy_adam", action="store_true", help="row-sparse Adam on the item table (a deviation, DESIGN.md section 8)")
    ap.add_argument("--sparse-exchange", default="auto", choices=["auto", "on", "off"], help="DP: item-table gradient exchange")
    ap.add_argument("--condition-ms", type=float, default=15.0, help="untimed steps of the workload run in front of the warm-up steps for this "
                    "many milliseconds (the clock governor's transient after idle, reported as `conditioning`); 0 = none")
    ap.add_argument("--profile-json", default=None, help="write the per-kernel HIP-event table here")
    ap.add_argument("--launcher-selftest", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # this process stays off the GPU: N fresh rank processes of this script, rank 0 prints the JSON line
        sys.exit(launch_ranks(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:]))
    if args.launcher_selftest:
        return launcher_selftest()

    import numpy as np
    import torch
    import castrec_amd  # noqa: F401
    from castrec_amd import dist as D_
    from castrec_amd import engine as E
    from castrec_amd import synth
    from castrec_amd.sampler import WarpSampler
    globals().update(np=np, torch=torch, E=E)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
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
    corpus = synth.preset(args.corpus)
    sargs = types.SimpleNamespace(seed=42, bin_in_hours=48, max_bins=200, log_scale=False)
    smp = WarpSampler(sargs, corpus, corpus.usernum, corpus.itemnum, batch_size=Bg, maxlen=T)
    NB = 16
    host_batches = []
    for _ in range(NB):
        u, seq, pos, neg, ts, rat, hrs, dys, _ = smp.next_batch()
        sl = slice(rank * B, (rank + 1) * B)
        host_batches.append((seq[sl], pos[sl], neg[sl], ts[sl], hrs[sl], dys[sl]))
    smp.close()
    # (the batches are packed per engine: a slot = the six id rows + the batch's occurrence index, Engine.pack_slot)

    conditioning = []

    def run(precision, steps, warmup):
        eng = E.Engine(args.model, corpus.usernum, corpus.itemnum, hyper(args), B, training=True, n_slabs=args.n_slabs,
                       batch_global=Bg, row_offset=rank * B * T, attn_precision=precision, lazy_adam=args.lazy_adam)
        dp = None
        use_graph = not args.no_graph
        staged = torch.from_numpy(np.stack([eng.pack_slot(*hb) for hb in host_batches])).cuda()
        run.staged = staged
        ring = (dist is None or not args.lazy_adam) and os.environ.get("CASTREC_NO_ID_RING") != "1"
        if ring:
            # the NB staged batches ARE the id ring: a step ends by moving the next step's batch (slot = step number mod NB) into
            # the static id buffers with extra workgroups of its Adam launch (Engine.use_id_ring; set before any capture: the
            # ring's address travels in that launch's arguments) -- no host-issued copy between two steps
            eng.use_id_ring(staged)
        if dist is not None:
            # data-parallel step: three HIP graphs (forward + backward up to the last table-gradient launch | rest of the backward
            # + slab collapse | Adam) around the collectives; the table's exchange runs beside the second graph (dist.step_phases)
            eng.load_slot(staged[0])
            rep = D_.EngineReplica(eng, use_graph=use_graph)
            dp = D_.DataParallel(rep, rank, world, sparse={"auto": None, "on": True, "off": False}[args.sparse_exchange],
                                 force_collectives=force_dist)
            if use_graph:
                # one HIP graph for the whole step INCLUDING the collectives (one rank, CASTREC_FORCE_DIST=1: +3.3 % over the plain step
                # against +12.8 % for three graphs and eager collectives): the default at every rank count since round 4 -- with more
                # than one rank the capture is followed by one replayed step whose result must leave the replicas identical
                # (dist.DataParallel._replayed_step_agrees), else the three-graph form runs; CASTREC_DP_ONE_GRAPH=0 forces that form
                whole = dp.capture_step() if os.environ.get("CASTREC_DP_ONE_GRAPH") != "0" else False
                eng.set_step(1); eng.Mom.zero_(); eng.Vel.zero_(); eng.Gflat.zero_()
        # steps per graph launch, as the fed training path runs them (models.Model.steps_per_launch: 4 where four batches wait in the id ring):
        # the device idles 5-9 us between two graph launches; here the batches of all steps are resident, so every launch is a full one
        spg = max(1, int(os.environ.get("CASTREC_STEPS_PER_GRAPH", "4"))) if (ring and dist is None and use_graph and not args.lazy_adam) else 1
        run.spg = spg
        if dist is None and use_graph:
            eng.load_slot(staged[0])
            eng.capture(n_steps=spg)
            eng.set_step(1); eng.Mom.zero_(); eng.Vel.zero_(); eng.Gflat.zero_()
        if ring:
            eng.load_slot(staged[eng.step_number() % NB])

        def step(i):
            if not ring:
                eng.load_slot(staged[i % NB])
            if dp is None:
                if use_graph:
                    eng.graph.launch()
                else:
                    eng.launch_step()
            elif use_graph:
                dp.step_phases()                             # graphs + RCCL over xGMI: table exchange beside the rest of the backward
            else:
                eng.launch_backward_to_flat()
                dp.exchange(eng.Gflat)                       # dense bucket all-reduce, or sparse table rows + small bucket
                eng.launch_adam_from_flat()

        # Conditioning (untimed, reported): after ANY idle period the chip needs ~35 steps (13 ms) of this workload before its step
        # time settles -- 0.390 ms for the first steps, 0.372 from about the 35th on, whatever ran before (30 ms of streaming
        # kernels do not shorten it; tools/probes/step_transient.py) -- so a window of W = 5 + K = 20 steps right behind the
        # setup measures the governor's transient, not the training rate.  `--condition-ms` (default 15) of the same steps run
        # first; their own mean time goes into the line as `cold_ms_per_step`.  0 turns it off.
        cond = {"steps": 0, "cold_ms_per_step": None}
        if args.condition_ms > 0:
            torch.cuda.synchronize()
            tc = time.perf_counter()
            nc = 0
            # (with collectives every rank must run the SAME number of steps: a fixed 250 -- RCCL's lazy set-up of its channels
            #  skews the first ~200 exchanges)
            while (nc < 250) if dist is not None else (nc < 8 or (time.perf_counter() - tc) * 1e3 < args.condition_ms):
                step(nc)
                nc += 1
                if nc % 8 == 0:
                    torch.cuda.synchronize()
            torch.cuda.synchronize()
            cond = {"steps": nc, "cold_ms_per_step": round((time.perf_counter() - tc) / nc * 1e3, 4)}
        conditioning.append(cond)
        def run_steps(n, i0):
            # EXACTLY n steps: whole graphs of spg steps, the rest one step per launch
            if spg > 1:
                for _ in range(n // spg):
                    eng.graph_multi.launch()
                for i in range(n % spg):
                    step(i0 + i)
            else:
                for i in range(n):
                    step(i0 + i)

        if spg > 1:
            eng.graph_multi.launch()                         # (untimed: a graph's first launch uploads it)
        run_steps(warmup, 0)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(steps, warmup)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt[0])
        return eng, dp, dt, use_graph

    eng, dp, dt, use_graph = run(args.attn_precision, args.steps, args.warmup)
    loss, auc = eng.loss_auc()
    prec = eng.attn_precision

    if rank == 0:
        per_launch, by_name = kernel_profile(eng, run.staged)
        dom = max(by_name.items(), key=lambda kv: kv[1]["us"])
        name, d = dom
        bound = d["bound"]
        peak, unit = PEAK[bound]
        if bound == "mfma":
            achieved = d["flops"] / (d["us"] * 1e-6) / 1e12
        else:
            achieved = d["bytes"] / (d["us"] * 1e-6) / 1e9
        traffic, mfma_busy, src = pmc_lookup(name, prec, config_tag(args))
        roofline = dict(kernel=name, bound=bound, achieved=round(achieved, 3), peak=peak, unit=unit,
                        frac=round(achieved / peak, 5), traffic=traffic, launches_per_step=d["launches"],
                        us_per_launch=round(d["us"] / d["launches"], 2),
                        algorithmic_per_launch=(d["flops"] if bound == "mfma" else d["bytes"]) / d["launches"],
                        mfma_busy_frac=mfma_busy,
                        note="achieved = algorithmic %s per launch / average HIP-event duration of its %d launches per step "
                             "(eager instrumented pass, 8 steps); peak = fp32-input MFMA (the arithmetic is fp32-grade: f32 MFMA, or bf16 "
                             "MFMA on hi+lo split operands); traffic (HBM bytes per launch, (2*FETCH_SIZE + WRITE_SIZE) KiB) and "
                             "mfma_busy_frac (matrix-pipe busy share) are NOT measured in this run: they are read from the committed "
                             "rocprofv3 --pmc summary of this command and workload: %s"
                             % ("flops" if bound == "mfma" else "bytes", d["launches"], src))
        if name.startswith("cr_attn") or name in ("cr_stack_fwd", "cr_stack_block_bwd"):
            roofline["executed_flop_frac"] = executed_tile_fraction(host_batches, T)
        # the attention entries on their own (the kernels the north-star target is stated on), whichever entry dominates
        attn = {}
        for n_ in ("cr_attn_fwd", "cr_attn_bwd", "cr_stack_fwd", "cr_stack_block_bwd"):    # (the last two hold the attention core since rounds 2 / 3)
            if n_ in by_name:
                a = by_name[n_]
                tf = a["flops"] / (a["us"] * 1e-6) / 1e12
                tr, mb, _ = pmc_lookup(n_, prec, config_tag(args))
                attn[n_] = dict(us_per_launch=round(a["us"] / a["launches"], 2), launches_per_step=a["launches"], achieved_TFLOPs=round(tf, 2),
                                frac_of_f32_mfma_peak=round(tf / 157.3, 4), frac_of_bf16_mfma_peak=round(tf / PEAK_BF16_TF, 5),
                                mfma_busy_frac=mb, traffic=tr)
        # every entry point of the step: time per step, launches, and the algorithmic rate against its bound
        kernels = {}
        for n_, a in sorted(by_name.items(), key=lambda kv: -kv[1]["us"]):
            pk, un = PEAK[a["bound"]]
            rate = (a["flops"] / 1e12 if a["bound"] == "mfma" else a["bytes"] / 1e9) / (a["us"] * 1e-6) if a["us"] > 0 else 0.0
            kernels[n_] = dict(us_per_step=round(a["us"], 1), launches=a["launches"], bound=a["bound"],
                               achieved=round(rate, 2) if (a["flops"] or a["bytes"]) else None, unit=un)
        attn["executed_flop_frac"] = executed_tile_fraction(host_batches, T)
        attn["arithmetic"] = {"f32": "v_mfma_f32_16x16x4_f32", "bf16x3": "v_mfma_f32_16x16x32_bf16 x3 (hi*hi + hi*lo + lo*hi), fp32 accumulate",
                              "bf16": "v_mfma_f32_16x16x32_bf16, fp32 accumulate"}[prec]
        if args.profile_json:
            with open(args.profile_json, "w") as f:
                json.dump(dict(per_launch=per_launch, by_name=by_name, n_launches=eng.n_launches()), f, indent=1, default=str)
        cfg = {"workload": "%s-shaped synthetic (%d users, %d items), %s maxlen=%d hidden_units=%d num_blocks=%d "
                           "num_heads=%d dropout=%.2f, batch %d/GPU (global %d), fwd+bwd+%s TF-Adam per step"
                           % (args.corpus, corpus.usernum, corpus.itemnum, args.model, T, args.hidden_units, args.num_blocks, args.num_heads,
                              args.dropout_rate, B, Bg, "row-sparse (lazy)" if args.lazy_adam else "dense"),
               "parallelism": "dp%d" % world, "hip_graph": use_graph, "steps_per_graph_launch": getattr(run, "spg", 1), "launches_per_step": eng.n_kernel_launches(), "abi_calls_per_step": eng.n_launches(),
               "attn_precision": prec, "final_loss": round(loss, 5), "final_auc": round(auc, 5)}
        if dist is not None:
            cfg["collective"] = {"backend": dist.get_backend(), "ranks": dist.get_world_size(),
                                 "step": "one HIP graph incl. the collectives" if dp._step_graph is not None else "three HIP graphs, collectives between them (table exchange beside the rest of the backward)",
                                 "table_exchange": "sparse rows (all-gather)" if dp.sparse else "dense (in the bucket all-reduce)",
                                 "bucket_floats": int(eng.Gflat.numel())}
        out = {
            "metric": "user-sequences/sec (fwd+bwd)", "value": round(Bg * args.steps / dt, 1), "unit": "sequences/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "conditioning": dict(conditioning[0], note="untimed steps of the same workload in front of the W warm-up steps (--condition-ms): the chip's "
                                 "step time settles ~35 steps after any idle period; cold_ms_per_step = the mean over these first steps"),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"f32": "f32", "bf16x3": "f32 (bf16x3 split MFMA in attention)", "bf16": "bf16 (attention) / f32"}[prec],
            "data": "synthetic", "config": cfg, "roofline": roofline, "attention": attn, "kernels": kernels,
        }
    del eng
    torch.cuda.empty_cache()
    if rank == 0:
        if world == 1 and not args.no_extra_precisions:
            # the same step in the other attention arithmetics (short runs; the headline line above is `attn_precision`)
            others = {}
            for p_ in ("f32", "bf16x3", "bf16"):
                if p_ == prec:
                    continue
                e2, _, dt2, _ = run(p_, max(20, args.steps // 4), 10)
                others[p_] = dict(value=round(B * max(20, args.steps // 4) / dt2, 1), ms_per_step=round(dt2 / max(20, args.steps // 4) * 1e3, 4))
                del e2
                torch.cuda.empty_cache()
            out["other_precisions"] = others
        if world == 1 and not args.no_other_configs:
            out["other_configs"] = other_configs()
        if world == 1 and not args.no_gather:
            out["gather"] = gather_block()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.model, T, args, host_batches, corpus.usernum, corpus.itemnum)
            c1 = cpu_baseline("sasrec", 50, args, host_batches, corpus.usernum, corpus.itemnum, budget_s=8.0)   # BASELINE configs[0]
            c1["config"] = "configs[0]: ml-1m SASRec maxlen=50 hidden_units=50 num_blocks=2 num_heads=1 batch=128"
            out["cpu_baseline_c1"] = c1
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

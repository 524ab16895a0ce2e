"""End-to-end on the GPU through the reference-shaped surface: Model classes, WarpSampler, evaluate,
the main.py CLI loop, and the data-parallel step path (single rank)."""
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def chain_corpus(n_users=200, n_items=60, length=30, seed=0):
    """Perfectly sequential data: item i is always followed by i+1 (mod n_items) -> HR@10 can reach 1."""
    import castrec_amd  # noqa: F401
    from castrec_amd import synth
    rs = np.random.RandomState(seed)
    d = {}
    for u in range(1, n_users + 1):
        s = rs.randint(0, n_items)
        d[u] = [((s + k) % n_items + 1, 4.0, 1_000_000_000 + 86400 * k + u) for k in range(length)]
    return synth.from_dict(d, n_users, n_items)


@pytest.mark.parametrize("D", [32, 128])          # the register-layout kernels / the wide row kernels (cr_wide.hip, eval: unfused chain sizes too)
def test_sasrec_learns_chain_and_evaluate_reports_high_hr(D):
    import castrec_amd  # noqa: F401
    from castrec_amd.models import SASRec
    from castrec_amd.sampler import WarpSampler
    from castrec_amd import util as U
    c = chain_corpus()
    dataset = U.partition(c.to_dict(), c.usernum, c.itemnum)
    args = types.SimpleNamespace(maxlen=20, hidden_units=D, num_blocks=2, num_heads=2, dropout_rate=0.1, l2_emb=0.0, lr=3e-3 if D == 32 else 1e-3,
                                 max_bins=20, num_context_blocks=1, seed=5, bin_in_hours=24, log_scale=False,
                                 test_model=None, test_seq_len=None)
    np.random.seed(5)
    model = SASRec(c.usernum, c.itemnum, args)
    smp = WarpSampler(args, U.train_corpus(dataset[0], c.usernum, c.itemnum), c.usernum, c.itemnum, batch_size=64, maxlen=20)
    first = None
    for step in range(300):
        u, seq, pos, neg, ts, rat, hrs, dys, _ = smp.next_batch()
        out = model.train_step(u, seq, pos, neg, ts, hrs, dys, fetch=(step % 50 == 0 or step == 299))
        if first is None:
            first = out
    smp.close()
    auc, loss = out
    assert loss < 0.5 * first[1] and auc > 0.95
    ndcg, hr = U.evaluate(model, dataset, args)
    ndcg_v, hr_v = U.evaluate_valid(model, dataset, args)
    assert hr > 0.9 and hr_v > 0.9 and 0 < ndcg <= hr
    # predict() surface: reference-style single user with a shared candidate list
    logits, attn = model.predict(None, [1], [seq[0]], list(range(1, 102 if c.itemnum >= 101 else c.itemnum + 1)))
    assert logits.shape[0] == 1 and attn.shape == (2, 20, 20)


def test_main_cli_trains_evaluates_and_writes_reference_artifacts(tmp_path, monkeypatch):
    import main as cli
    monkeypatch.chdir(tmp_path)
    rc = cli.main(["--dataset", "synthetic:tiny", "--train_dir", "t", "--model", "cast_3", "--maxlen", "12", "--batch_size", "4",
                   "--hidden_units", "16", "--num_epochs", "2", "--eval_every", "1", "--max_bins", "20"])
    assert rc == 0
    runs = os.listdir(tmp_path / "saved_models" / "synthetic_tiny")
    assert len(runs) == 1
    d = tmp_path / "saved_models" / "synthetic_tiny" / runs[0]
    assert (d / "params.txt").exists() and (d / "model.ckpt").exists()
    lines = (d / "log.txt").read_text().strip().splitlines()
    assert len(lines) == 2 and lines[0].startswith("(")            # "(ndcg, hr) (ndcg, hr)" per evaluation (main.py:238)
    # TensorBoard scalars of the run directory (main.py:203,222-224,240-249): per epoch the training pair, then the evaluation
    from castrec_amd.tb_events import read_events
    evf = [x for x in os.listdir(d) if x.startswith("events.out.tfevents.")]
    assert len(evf) == 1
    ev = read_events(str(d / evf[0]))
    assert [s for s, _ in ev] == [1, 1, 2, 2] and set(ev[0][1]) == {"TRAIN/loss", "TRAIN/auc"}
    assert set(ev[1][1]) == {"VALID/NDCG@10", "VALID/HR@10", "TEST/NDCG@10", "TEST/HR@10"}
    import ast
    valid2, test2 = (ast.literal_eval(x) for x in lines[1].replace(") (", ")|(").split("|"))
    assert ev[3][1]["TEST/NDCG@10"] == pytest.approx(test2[0], rel=1e-6) and ev[3][1]["VALID/HR@10"] == pytest.approx(valid2[1], rel=1e-6)
    # --test_model mode re-loads the checkpoint (main.py:161-189)
    rc = cli.main(["--dataset", "synthetic:tiny", "--train_dir", "t", "--model", "cast_3", "--maxlen", "12", "--batch_size", "4",
                   "--hidden_units", "16", "--max_bins", "20", "--test_model", str(d), "--test_seq_len", "5"])
    assert rc == 0 and (d / "test_seq_len.txt").read_text().startswith("5,")
    assert (d / "attention_weights.svg").exists() and (d / "attention_weights.npy").exists()     # util.py:334-336


@pytest.mark.parametrize("D,H,n_slabs", [(20, 1, 8),          # fused D <= 64 kernels
                                         (128, 4, 8),         # cr_wide kernels that form the weight gradients themselves
                                         (256, 4, 40)])       # cr_wide + cr_gemm_wgrad on 32 of the 40 slabs: per-block slab counts
def test_dp_replica_single_rank_equals_plain_step(D, H, n_slabs):
    """the data-parallel step path (graph {forward, backward, cr_reduce_slabs}, exchange, Adam from the flat bucket) against the
    one-launch-chain step, on each family of row kernels"""
    import castrec_amd  # noqa: F401
    from castrec_amd import engine as E
    from castrec_amd.dist import DataParallel, EngineReplica
    rs = np.random.RandomState(1)
    B, T, itemnum = 8, 16, 50
    hp = E.Hyper(maxlen=T, hidden_units=D, num_blocks=1, num_heads=H, dropout_rate=0.2, max_bins=10, seed=2)
    a = E.Engine("cast_1", 9, itemnum, hp, B, training=True, n_slabs=n_slabs)
    b = E.Engine("cast_1", 9, itemnum, hp, B, training=True, n_slabs=n_slabs)
    if D == 256:
        assert a.n_wslabs == 32 < a.n_slabs and a.slab_counts is not None and int(a.slab_counts.min()) == 32
    b.P.copy_(a.P)
    dp = DataParallel(EngineReplica(b, use_graph=True), 0, 1)
    seq = rs.randint(1, itemnum + 1, (B, T)); seq[:, :4] = 0
    pos = rs.randint(1, itemnum + 1, (B, T)) * (seq != 0); neg = rs.randint(1, itemnum + 1, (B, T)) * (seq != 0)
    time = rs.randint(0, 11, (B, T)) * (seq != 0); z = np.zeros_like(seq)
    for _ in range(2):
        a.train_step(seq, pos, neg, time, z, z)
        dp.step((seq, pos, neg, time, z, z))
    torch.cuda.synchronize()
    pa, pb = a.get_params(), b.get_params()
    for k in pa:
        if k.endswith(".bk"):
            continue        # zero-gradient direction (softmax shift invariance): Adam amplifies rounding noise to O(lr)
        dlt = (pa[k] - pb[k]).abs()
        # (both paths add the same slabs, the plain step inside cr_adam_step and this one in cr_reduce_slabs: equal up to the
        #  rare element whose gradient nearly cancels and whose Adam move is then decided by the last bit)
        if D > 64:
            # (at these random starts the wide models have bias-like gradients of 1e7 ... 1e15 and many elements on an Adam sign
            #  boundary; the table's float atomics make the two engines' runs differ in the last bit: see test_dist_gpu.py)
            q99 = float(torch.quantile(dlt.flatten().float(), 0.99))
            assert q99 <= 1e-4 and float(dlt.max()) < 2.5e-3, (k, q99, float(dlt.max()))
        else:
            assert float(dlt.max()) <= 2e-6 or (float((dlt > 2e-6).float().mean()) < 1e-3 and float(dlt.max()) < 2.5e-3), (k, float(dlt.max()), float((dlt > 2e-6).float().mean()))
    assert a.loss_auc()[0] == pytest.approx(b.loss_auc()[0], rel=1e-5)


def test_model_loads_a_reference_tensorflow_checkpoint(tmp_path):
    """Model.load_tf_checkpoint on a bundle rebuilt from the reference's own index file (tests/golden/tf_index) and
    its trained cast_1 variables: every parameter ends up where the kernels read it; a wrong model class is refused."""
    import shutil
    import castrec_amd  # noqa: F401
    from castrec_amd import tf_bundle as tfb
    from castrec_amd.models import build_model
    here = os.path.dirname(os.path.abspath(__file__))
    idx_path = os.path.join(here, "golden", "tf_index", "cast_1.index")
    idx = tfb.read_index(idx_path)
    w = np.load(os.path.join(here, "golden", "cast_1_ml1m_weights.npz"))
    raw = np.zeros(max(e["offset"] + e["size"] for e in idx.values()) // 4, dtype="<f4")
    for name, e in idx.items():
        ln = tfb.logical_name(name)
        if ln is not None:
            raw[e["offset"] // 4:e["offset"] // 4 + w[ln].size] = w[ln].reshape(-1)
    prefix = str(tmp_path / "model.ckpt")
    shutil.copyfile(idx_path, prefix + ".index")
    raw.tofile(prefix + ".data-00000-of-00001")
    args = types.SimpleNamespace(maxlen=200, hidden_units=50, num_blocks=2, num_heads=1, dropout_rate=0.2, l2_emb=0.0, lr=1e-3,
                                 max_bins=200, num_context_blocks=2, seed=1, bin_in_hours=48, log_scale=False,
                                 test_model=None, test_seq_len=None, model="cast_1", batch_size=4)
    model = build_model("cast_1", 6040, 3416, 5, args)
    model.load_tf_checkpoint(prefix)
    got = model.get_params()
    assert set(got) == set(w.files)
    for k in w.files:
        assert np.array_equal(got[k].cpu().numpy(), w[k]), k
    other = build_model("sasrec", 6040, 3416, 5, args)
    with pytest.raises(ValueError, match="does not match model"):
        other.load_tf_checkpoint(prefix)


def test_checkpoint_restores_the_optimiser_state(tmp_path):
    """save -> a NEW model -> load -> train_step must equal continuing the original model: Adam's m / v and the step
    number (bias correction, dropout keys) travel with the parameters like tf.train.Saver's slots (main.py:159-175).
    The state is loaded BEFORE the new model's training engine exists -- the order main.py --test_model uses."""
    import types
    from castrec_amd import models
    rs = np.random.RandomState(21)
    B, T, itemnum = 6, 16, 40
    args = types.SimpleNamespace(maxlen=T, hidden_units=16, num_blocks=1, num_heads=1, dropout_rate=0.2, l2_emb=0.0, lr=1e-2,
                                 max_bins=8, num_context_blocks=1, seed=5)

    def batch(i):
        r = np.random.RandomState(50 + i)
        seq = r.randint(1, itemnum + 1, (B, T)); pos = r.randint(1, itemnum + 1, (B, T)); neg = r.randint(1, itemnum + 1, (B, T))
        seq[:, :3] = 0; pos[:, :3] = 0; neg[:, :3] = 0
        tm = r.randint(0, 9, (B, T)) * (seq != 0)
        z = np.zeros_like(seq)
        return None, seq, pos, neg, tm, z, z
    a = models.CAST1(9, itemnum, 5, args)
    for i in range(3):
        a.train_step(*batch(i))
    path = a.save(str(tmp_path / "ck.pt"))
    b = models.CAST1(9, itemnum, 5, args)
    b.load(path)                                         # no training engine yet: the slots wait for it
    ra = a.train_step(*batch(3))
    rb = b.train_step(*batch(3))
    torch.cuda.synchronize()
    assert ra[1] == pytest.approx(rb[1], rel=1e-6)       # same loss on the same batch: same parameters, same dropout key (step 4)
    pa, pb = a.get_params(), b.get_params()
    for k in pa:
        if k == "item_emb":                              # float atomics may reorder
            assert torch.allclose(pa[k], pb[k], rtol=0, atol=1e-6), k
        else:
            assert torch.equal(pa[k], pb[k]), k
    c = models.CAST1(9, itemnum, 5, args)                # without the slots the same step differs (fresh Adam at t = 1)
    c.load_params({k: v.cpu() for k, v in a.get_params().items()})
    assert a._train.step_number() == 5 and b._train.step_number() == 5


@pytest.mark.timeout(600)
def test_main_cli_two_ranks_equal_one_process(tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 2 main.py ...` (both ranks on the box's one card, bucket over gloo
    through host memory: CASTREC_DIST_BACKEND=gloo) ends with the numbers of the one-process run on the same GLOBAL batch:
    same sampler stream, row shards, global-row dropout keys, global target count."""
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    argv = [os.path.join(root, "main.py"), "--dataset", "synthetic:tiny", "--train_dir", "t", "--model", "cast_1", "--maxlen", "12",
            "--batch_size", "4", "--hidden_units", "16", "--num_epochs", "3", "--eval_every", "3", "--max_bins", "20"]
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    one, two = tmp_path / "one", tmp_path / "two"
    one.mkdir(); two.mkdir()
    r = subprocess.run([sys.executable] + argv, cwd=one, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port)] + argv, cwd=two, env=dict(env, CASTREC_DIST_BACKEND="gloo"),
                       capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]

    def last_log(d):
        runs = [x for x in os.listdir(d / "saved_models" / "synthetic_tiny")]
        assert len(runs) == 1
        run = d / "saved_models" / "synthetic_tiny" / runs[0]
        return run, (run / "log.txt").read_text().strip().splitlines()[-1]

    run1, l1 = last_log(one)
    run2, l2 = last_log(two)
    import re
    num = re.compile(r"(?<![\w.])\d+\.\d+(?:e-?\d+)?")           # "(np.float64(0.31), ...)": the decimal numbers, not the 64
    v1 = [float(x) for x in num.findall(l1)]
    v2 = [float(x) for x in num.findall(l2)]
    assert len(v1) == 4 and np.allclose(v1, v2, atol=2e-3), (l1, l2)             # NDCG / HR of valid and test
    assert (run2 / "model.ckpt").exists() and (run2 / "rank1" / "params.txt").exists()
    # the trained parameters agree to rounding (the two runs add the same shard gradients in a different order)
    a = torch.load(run1 / "model.ckpt", map_location="cpu", weights_only=False)
    b = torch.load(run2 / "model.ckpt", map_location="cpu", weights_only=False)
    assert a["P"].numel() == b["P"].numel() > 100
    # (elements whose gradient nearly cancels -- b_k above all -- take O(lr) Adam moves decided by rounding, in any
    #  implementation: DESIGN.md section 2; here the 4 x 16 key biases of the four blocks, 0.9 % of the vector)
    dp = np.abs(a["P"].numpy() - b["P"].numpy())
    assert (dp > 5e-5).mean() < 1.2e-2 and dp.max() < 5e-3, ((dp > 5e-5).mean(), dp.max())
    dm = np.abs(a["M"].numpy() - b["M"].numpy()).max()
    assert dm <= 1e-3 * np.abs(a["M"].numpy()).max(), (dm, np.abs(a["M"].numpy()).max())     # first moments, to their own scale

#!/usr/bin/env python3
"""Command-line training / evaluation loop with the flags of the reference's main.py:45-82.

    python main.py --dataset data/ml-1m.txt --train_dir sasrec_baseline --model sasrec --maxlen 200 --dropout_rate 0.2

Artefacts keep the reference's formats: saved_models/<dataset>/<train_dir>_<timestamp>/{params.txt,log.txt,model.ckpt}
(main.py:153-158,196-199,238).  `--dataset synthetic:<preset>` trains on a seeded synthetic corpus
(castrec_amd.synth.PRESETS) when no dataset file is available."""
import argparse
import json
import logging
import os
import random
import sys
import time
from datetime import datetime

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import castrec_amd  # noqa: E402
from castrec_amd import synth  # noqa: E402
from castrec_amd.engine import MODELS  # noqa: E402
from castrec_amd.models import build_model  # noqa: E402
from castrec_amd.sampler import WarpSampler  # noqa: E402
from castrec_amd.tb_events import EventWriter  # noqa: E402
from castrec_amd.util import data_partition, evaluate, evaluate_valid, partition, train_corpus  # noqa: E402


def parse_args(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--dataset', required=True, help='Location of pre-processed dataset (or synthetic:<preset>)')
    parser.add_argument('--maxlen', default=50, type=int)
    parser.add_argument('--train_dir', required=True)
    parser.add_argument('--batch_size', default=128, type=int)
    parser.add_argument('--lr', type=float, default=1e-3)
    parser.add_argument('--num_epochs', type=int, default=201)
    parser.add_argument('--max_norm', type=float, default=5.0)
    parser.add_argument('--hidden_units', default=50, type=int)
    parser.add_argument('--num_blocks', default=2, type=int)
    parser.add_argument('--num_heads', default=1, type=int)
    parser.add_argument('--dropout_rate', default=0.5, type=float)
    parser.add_argument('--l2_emb', default=0.0, type=float)
    parser.add_argument('--bin_in_hours', default=24, type=int)
    parser.add_argument('--max_bins', default=200, type=int)
    parser.add_argument('--num_context_blocks', default=2, type=int)
    parser.add_argument('--test_model', type=str, default=None)
    parser.add_argument('--test_seq_len', type=int, default=None)
    parser.add_argument('--saved_model', default='model.pt', type=str)
    parser.add_argument('--seed', default=42, type=int)
    parser.add_argument('--log_scale', type=bool, default=False)       # type=bool as in the reference (any string is True)
    parser.add_argument('--input_context', type=bool, default=False)
    parser.add_argument('--model', default="cast_1", required=True, help="model to use from" + str(MODELS))
    parser.add_argument('--eval_every', type=int, default=20, help='evaluate + checkpoint every N epochs (reference: 20)')
    return parser.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    logger = logging.getLogger('ir2')
    logging.basicConfig(level=logging.INFO, format="%(asctime)s [%(levelname)-5.5s]  %(message)s",
                        handlers=[logging.FileHandler(os.path.join('.', 'output.log')), logging.StreamHandler()])
    if args.dataset.startswith("synthetic:"):
        c = synth.preset(args.dataset.split(":", 1)[1])
        dataset = partition(c.to_dict(), c.usernum, c.itemnum)
    elif not os.path.exists(args.dataset):
        logger.info('Pre-process the data first')
        sys.exit()
    else:
        dataset = data_partition(args.dataset, args.log_scale)
    train, valid, test, usernum, itemnum, ratingnum = dataset
    num_batch = round(len(train) / args.batch_size)               # main.py:96 (banker's rounding)
    print('usernum', usernum, 'itemnum', itemnum)
    cc = sum(len(v) for v in train.values())
    logger.info('Average sequence length: {:.2f}'.format(cc / len(train)))
    if args.seed:                                                  # main.py:103-107
        random.seed(args.seed)
        np.random.seed(args.seed)
    if args.model.lower() not in MODELS:
        print("provide model from", MODELS)
        sys.exit(0)
    # Data parallelism (not in the reference): under `python -m torch.distributed.run --nproc-per-node N
    # main.py ...` every rank runs this same program -- same seed, same sampler stream, --batch_size is the GLOBAL batch --
    # and trains on its rows of each batch; gradients meet over RCCL (castrec_amd.dist).  Rank 0 evaluates, logs and saves.
    # The process group comes FIRST: init_from_env selects this rank's GPU (torch.cuda.set_device(LOCAL_RANK)), and the
    # model's parameter vector must be allocated on that device, not on cuda:0 of every rank.
    rank, world = 0, 1
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        from castrec_amd import dist as cr_dist
        import torch.distributed as tdist
        rank, _, world = cr_dist.init_from_env(os.environ.get("CASTREC_DIST_BACKEND"))
    model = build_model(args.model, usernum, itemnum, ratingnum, args)
    if world > 1:
        model.data_parallel(rank, world)
    sampler = WarpSampler(args, train_corpus(train, usernum, itemnum), usernum, itemnum,
                          batch_size=args.batch_size, maxlen=args.maxlen, n_workers=1)
    model_path = os.path.abspath('saved_models')
    now = datetime.now()
    files_path = os.path.join(model_path, os.path.basename(args.dataset).replace(":", "_"),
                              '{}_{}'.format(args.train_dir, now.strftime("%m-%d-%Y-%H-%M-%S")))
    save_path = os.path.join(files_path, 'model.ckpt')

    if args.test_model:                                            # main.py:161-189
        ck = os.path.join(args.test_model, 'model.ckpt')
        if os.path.exists(ck) or os.path.exists(ck + '.index'):
            if os.path.exists(ck + '.index'):
                model.load_tf_checkpoint(ck)                            # a run directory written by the reference (TF bundle)
            else:
                model.load(ck)
            print('loaded saved model {}'.format(args.test_model))
            u, seq, pos, neg, timeseq, ratings_seq, hours_seq, days_seq, _ = sampler.next_batch()
            auc, loss = model.train_step(u, seq, pos, neg, timeseq, hours_seq, days_seq)   # the reference's one-train-step quirk
            print(auc); print(loss)
            t_test = evaluate(model, dataset, args)
            logger.info('test (NDCG@10: %.4f, HR@10: %.4f)' % (t_test[0], t_test[1]))
            with open(os.path.join(args.test_model, 'test_seq_len.txt'), 'a') as f:
                f.write('{},{},{}\n'.format(args.test_seq_len, t_test[0], t_test[1]))
        else:
            print('{} not found'.format(args.test_model))
        sampler.close()
        return 0

    if rank != 0:
        files_path = os.path.join(files_path, 'rank%d' % rank)     # (the other ranks write nothing of interest)
    os.makedirs(files_path, exist_ok=True)
    with open(os.path.join(files_path, 'params.txt'), 'w') as f:   # main.py:196-197
        json.dump(args.__dict__, f, indent=2)
    f = open(os.path.join(files_path, 'log.txt'), 'w')
    # TensorBoard scalars as the reference's tf.summary.FileWriter(TRAIN_FILES_PATH) leaves them (main.py:203,222-224,240-249;
    # sasrec.py:112-118): TRAIN/loss, TRAIN/auc of each epoch's last step; VALID/* and TEST/* at every evaluation
    writer = EventWriter(files_path)
    T, t0, rc = 0.0, time.time(), 0
    def hand_over():
        # the sampler's next batch goes to the device while the steps before it run (model.feed: one pinned copy on a copy
        # stream); the reference feeds each step through sess.run's feed_dict (main.py:212-219) -- same batches, same order
        u, seq, pos, neg, timeseq, ratings_seq, hours_seq, days_seq, _ = sampler.next_batch()
        model.feed(u, seq, pos, neg, timeseq, hours_seq, days_seq)

    try:
        total, done, out = args.num_epochs * num_batch, 0, None
        # how many handed-over batches may wait: 1 (the step that runs now has its successor's batch in place), or -- a model whose
        # graph launches run several steps each (Model.steps_per_launch) -- that many plus one
        ahead = int(getattr(model, "feed_ahead", 1))
        fed = 0
        if total > 0:
            hand_over(); fed = 1
        for epoch in range(1, args.num_epochs + 1):
            if ahead <= 1:
                for step in range(num_batch):
                    done += 1
                    if done < total:
                        hand_over(); fed += 1                      # one batch ahead of the step that runs now
                    last = (step == num_batch - 1)
                    out = model.train_fed(fetch=last)
            else:
                end = epoch * num_batch
                while done < end:
                    while fed < total and fed - done < ahead:
                        hand_over(); fed += 1
                    done += model.train_fed_many(max_steps=end - done)     # (never across the epoch's end: its last step's loss is logged)
                out = model.loss_auc()
            if out is not None:
                logger.info('epoch %d: TRAIN/loss %.5f TRAIN/auc %.5f' % (epoch, out[1], out[0]))
                writer.add_scalars(epoch, {'TRAIN/loss': out[1], 'TRAIN/auc': out[0]})
                writer.flush()
            if epoch % args.eval_every == 0 and rank != 0:
                tdist.barrier()                                    # rank 0 evaluates; parameters are identical everywhere
                t0 = time.time()
            elif epoch % args.eval_every == 0:
                logger.info('Model saved in path: %s' % model.save(save_path))
                logger.info('Evaluating')
                T += time.time() - t0
                t_test = evaluate(model, dataset, args)
                t_valid = evaluate_valid(model, dataset, args)
                logger.info('epoch:%d, time: %f(s), valid (NDCG@10: %.4f, HR@10: %.4f), test (NDCG@10: %.4f, HR@10: %.4f)' % (
                    epoch, T, t_valid[0], t_valid[1], t_test[0], t_test[1]))
                f.write(str(tuple(float(x) for x in t_valid)) + ' ' + str(tuple(float(x) for x in t_test)) + '\n')   # plain floats, as main.py:238 prints under numpy 1.16
                f.flush()
                writer.add_scalars(epoch, {'VALID/NDCG@10': float(t_valid[0]), 'VALID/HR@10': float(t_valid[1]),
                                           'TEST/NDCG@10': float(t_test[0]), 'TEST/HR@10': float(t_test[1])})
                writer.flush()
                if world > 1:
                    tdist.barrier()
                t0 = time.time()
    except Exception as e:                                         # main.py:253-257
        logger.error(e)
        rc = 1
        if world > 1:
            # the other ranks sit in a barrier (or a collective) this rank will never reach: leave non-zero NOW so that the
            # launcher tears the peers down instead of letting them wait for the process-group timeout
            f.close()
            sampler.close()
            logging.shutdown()
            os._exit(1)
    f.close()
    writer.close()
    sampler.close()
    if rc == 0:
        print("Done")
    return rc


if __name__ == '__main__':
    sys.exit(main())

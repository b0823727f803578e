"""Training driver counterpart of /root/reference/code/train_dual.py:100-281 on synthetic meshes.

Same loop structure and flag names for the parts that touch the hot path -- gradient accumulation
over ``--batch_size`` meshes (here: one disjoint-union step), L1/L2 losses with scales, Adam / SGD /
RMSprop, the five LR schedules (step / multi_step / exp / auto / lambda), per-epoch evaluation with node-count-weighted means
(train_dual.py:233-263), best-on-eval checkpoint with the reference's state-dict keys.  The datasets
of the reference are external downloads, so meshes are noisy icospheres (meshgen.py).  Data parallel:
launch through ``python -m torch.distributed.run --nproc-per-node N tools/train_synthetic.py ...``;
meshes are sharded by rank and the flat gradient bucket is all-reduced once per step.

  python tools/train_synthetic.py --max_epoch 5 --batch_size 4 --freq 16 --n_train 12 --n_eval 4
"""
import argparse
import json
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geobi_gnn_amd import network, meshgen, train_util          # noqa: E402
from geobi_gnn_amd.data import union_batch_graphs, RandomRotate   # noqa: E402
from geobi_gnn_amd.parallel import init_distributed, FlatParameters, shard_indices, batched_losses   # noqa: E402


def parse_arguments(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('--freq', type=int, default=16, help='icosphere frequency (F = 20 freq^2)')
    p.add_argument('--n_train', type=int, default=12)
    p.add_argument('--n_eval', type=int, default=4)
    train_util.add_training_flags(p)                  # the reference's flag names (train_dual.py:57-82)
    p.set_defaults(max_epoch=10, batch_size=4)
    p.add_argument('--rotate', type=int, default=0, help='1: random z rotation per batch, 2: full 3-axis (dataset.py:39-69)')
    p.add_argument('--seed', type=int, default=40938661)
    p.add_argument('--out', type=str, default='')
    p.add_argument('--tb_dir', type=str, default='', help='TensorBoard event files: <tb_dir>/train (per iteration) and <tb_dir>/test '
                   '(per epoch), the scalars of train_dual.py:222-226,262-266 (train_util.SummaryWriter)')
    return p.parse_args(argv)


def main(argv=None):
    opt = parse_arguments(argv)
    rank, world, device = init_distributed()
    assert torch.cuda.is_available(), 'the geobi path runs on the MI355X only'
    torch.manual_seed(opt.seed)
    sigmas = (0.1, 0.2, 0.3)
    train = [meshgen.synthetic_dual_data(opt.freq, sigmas[i % 3], seed=1000 + i) for i in range(opt.n_train)]
    evals = [meshgen.synthetic_dual_data(opt.freq, sigmas[i % 3], seed=5000 + i) for i in range(opt.n_eval)]
    evals = [(a.to(device), b.to(device)) for a, b in evals]
    # training meshes live on the device with everything that depends on ONE mesh built once (adjacency, reverse-edge
    # index, vertex -> corner lists); a step's batch is two concatenation launches (data.union_batch_graphs)
    from geobi_gnn_amd.network import _fv_index
    train = [(a.to(device), b.to(device)) for a, b in train]
    for a, b in train:
        a.graph().ensure_in(); b.graph().ensure_in()
        _fv_index(b, a.x.shape[0])[1].get()

    net = network.DualGNN(force_depth=False, pool_type='max', wei_param=opt.wei_param).to(device)
    flat = FlatParameters(net)
    optimizer = train_util.make_optimizer(opt, flat.parameters(), fused=True if opt.optimizer == 'adam' else None)
    sch = train_util.make_scheduler(opt, optimizer)
    ckpt = train_util.BestCheckpoint(opt.out if rank == 0 else '')

    import numpy as np
    rotate = None if not opt.rotate else RandomRotate(z_rotated=opt.rotate == 1, rng=np.random.default_rng(opt.seed + rank))
    best, history = math.inf, []
    train_writer = test_writer = None
    if opt.tb_dir and rank == 0:
        import os
        train_writer = train_util.SummaryWriter(os.path.join(opt.tb_dir, 'train'))
        test_writer = train_util.SummaryWriter(os.path.join(opt.tb_dir, 'test'))
        test_writer.add_text('train_params', str(opt))
    iteration = 0
    for epoch in range(1, opt.max_epoch + 1):
        net.train()
        mine = shard_indices(len(train), rank, world, seed=opt.seed, epoch=epoch)
        t0 = time.time()
        pending = []
        for s in range(0, len(mine), opt.batch_size):
            dv, df = union_batch_graphs([train[i] for i in mine[s:s + opt.batch_size]])
            if rotate is not None:
                rotate((dv, df))
            flat.bucket.zero()
            vp, npred, _ = net((dv.shallow_copy(), df.shallow_copy()))
            lv, ln = batched_losses(vp, npred, dv, df, opt.loss_v, opt.loss_n)
            loss = network.dual_loss(lv, ln, opt.loss_v_scale, opt.loss_n_scale)
            loss.backward()
            flat.bucket.all_reduce_mean()
            optimizer.step()
            iteration += len(mine[s:s + opt.batch_size])
            if train_writer is not None:
                # the reference reads five scalars back per iteration (.item()); here they stay on the device until the
                # epoch ends -- one host read per epoch instead of one per step
                with torch.no_grad():
                    pending.append((iteration, torch.stack([lv.detach(), ln.detach(), loss.detach(),
                                                            network.error_v(vp.detach(), dv.y), network.error_n(npred.detach(), df.y)])))
        if train_writer is not None:
            for it, vals in pending:
                for tag, v in zip(('loss_v', 'loss_f', 'dual_loss', 'error_v', 'error_f'), vals.tolist()):
                    train_writer.add_scalar(tag, v, it)
            train_writer.flush()
        # evaluation: node-count-weighted means over the eval meshes of every rank (train_dual.py:233-259)
        net.eval()
        meter = train_util.EvalMeter()
        with torch.no_grad():
            for a, b in evals[rank::world]:
                vp, npred, _ = net((a.shallow_copy(), b.shallow_copy()))
                meter.add_prediction(vp, npred, a, b, opt.loss_v, opt.loss_n)
        res = meter.result(device)
        rec = {'epoch': epoch, 'train_loss': float(loss.detach()), 'eval_loss_v': res['eval_loss_v'],
               'eval_loss_f': res['eval_loss_f'], 'eval_error_v': res['eval_error_v'],
               'eval_error_f_deg': res['eval_error_f'], 'lr': optimizer.param_groups[0]['lr'],
               'epoch_s': round(time.time() - t0, 3)}
        history.append(rec)
        saved = ckpt.update(net, rec['eval_error_f_deg'])     # keys: gnn_v.l_conv1.lin.weight ... fc_f2.bias
        if rank == 0:
            print(json.dumps(dict(rec, saved=saved)), flush=True)
        if test_writer is not None:
            for tag, key in (('loss_v', 'eval_loss_v'), ('loss_f', 'eval_loss_f'), ('error_v', 'eval_error_v'), ('error_f', 'eval_error_f')):
                test_writer.add_scalar(tag, res[key], iteration)
            test_writer.flush()
        train_util.step_scheduler(opt, sch, rec['eval_error_f_deg'])
    for w in (train_writer, test_writer):
        if w is not None:
            w.close()
    return history


if __name__ == '__main__':
    main()

"""The pieces of the reference's training loop that touch the hot path's parameters and outputs, behind the same
flag names (/root/reference/code/train_dual.py:39-96 flags, :162-180 optimisers and
learning-rate schedules, :233-263 evaluation means, :270-276 best-on-eval checkpoint).

The loop itself (data loading, TensorBoard, progress bars) is the reference's Python harness and stays out of
scope; tools/train_synthetic.py drives these helpers on synthetic meshes.
"""
import math

import torch
from torch.optim import lr_scheduler

from . import network
from .parallel import reduce_sums

LR_SCHEDULES = ('step', 'multi_step', 'exp', 'auto', 'lmd')


def add_training_flags(parser):
    """The optimiser / schedule / loss flags of train_dual.py:57-82, same names, types and defaults."""
    parser.add_argument('--loss_v', type=str, default='L1')
    parser.add_argument('--loss_n', type=str, default='L1')
    parser.add_argument('--loss_v_scale', type=float, default=1)
    parser.add_argument('--loss_n_scale', type=float, default=1)
    parser.add_argument('--wei_param', type=int, default=2)
    parser.add_argument('--max_epoch', type=int, default=1000)
    parser.add_argument('--batch_size', type=int, default=1)
    parser.add_argument('--lr_sch', type=str, default='lmd')
    parser.add_argument('--lr', type=float, default=0.001)
    parser.add_argument('--lr_step', type=int, nargs='+', default=[10])
    parser.add_argument('--lr_decay', type=float, default=1)
    parser.add_argument('--optimizer', type=str, default='adam')
    parser.add_argument('--momentum', type=float, default=0.9)
    parser.add_argument('--beta1', type=float, default=0.9)
    parser.add_argument('--beta2', type=float, default=0.999)
    parser.add_argument('--weight_decay', type=float, default=0)
    return parser


class FlatAdam(torch.optim.Optimizer):
    """torch.optim.Adam (train_dual.py:162; no amsgrad, no maximize) for fp32 parameters on the MI355X, one
    `geobi_adam_step` launch per parameter tensor -- meant for `parallel.FlatParameters`, where the whole network is ONE
    tensor and its gradient the bucket the kernels write (torch's own fused Adam hands a workgroup 65 536 elements: 15
    workgroups, 45 us per step for the 0.94 M parameters; this launch ~6 us).  Same state layout as torch.optim.Adam
    (`step`, `exp_avg`, `exp_avg_sq` per parameter), so state dicts move between the two."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError('FlatAdam: lr %r betas %r eps %r weight_decay %r' % (lr, betas, eps, weight_decay))
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        from . import _lib as L
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            # a state dict of torch.optim.Adam carries its options along: the ones that change the update rule must not be
            # dropped silently (this kernel implements the plain rule the reference uses, train_dual.py:162)
            for flag in ('amsgrad', 'maximize'):
                if group.get(flag):
                    raise ValueError('FlatAdam implements plain Adam: the loaded options ask for %s=True '
                                     '(use torch.optim.Adam for that state dict)' % flag)
            b1, b2 = group['betas']
            for p in group['params']:
                g = p.grad
                if g is None:
                    continue
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and g.is_contiguous()
                        and g.dtype == torch.float32 and g.device == p.device):
                    raise L.GeobiError('FlatAdam: contiguous fp32 parameters and gradients on the MI355X only')
                st = self.state[p]
                if len(st) == 0:
                    st['step'] = torch.zeros((), dtype=torch.float32)
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st['step'] += 1
                t = float(st['step'])                     # host scalar (a state dict loaded from a fused torch Adam: one read)
                L.call('geobi_adam_step', L.ptr(p), L.ptr(g), L.ptr(st['exp_avg']), L.ptr(st['exp_avg_sq']), p.numel(),
                       float(group['lr']), float(b1), float(b2), float(group['eps']), float(group['weight_decay']),
                       1.0 - b1 ** t, 1.0 - b2 ** t, L.stream())
        return loss


def make_optimizer(opt, params, fused=None):
    """train_dual.py:162-167.  `fused`: single-launch Adam over the flat parameter (FlatAdam: same update rule, same
    state layout)."""
    if opt.optimizer == 'sgd':
        return torch.optim.SGD(params, lr=opt.lr, momentum=opt.momentum, weight_decay=opt.weight_decay)
    if opt.optimizer == 'rmsprop':
        return torch.optim.RMSprop(params, lr=opt.lr, alpha=0.9)
    if opt.optimizer == 'adam':
        if fused and torch.cuda.is_available():
            return FlatAdam(params, lr=opt.lr, betas=(opt.beta1, opt.beta2), weight_decay=opt.weight_decay)
        return torch.optim.Adam(params, lr=opt.lr, betas=(opt.beta1, opt.beta2), weight_decay=opt.weight_decay)
    raise ValueError('optimizer %r: the reference knows sgd, rmsprop and adam' % (opt.optimizer,))


def make_scheduler(opt, optimizer):
    """train_dual.py:169-180: 'step', 'multi_step', 'exp', 'auto' (reduce on plateau of the evaluation normal
    error), anything else = the exponential lambda rule lr * decay^(epoch / lr_step[0])."""
    steps = list(opt.lr_step) if isinstance(opt.lr_step, (list, tuple)) else [opt.lr_step]
    if opt.lr_sch == 'step':
        return lr_scheduler.StepLR(optimizer, step_size=steps[0], gamma=opt.lr_decay)
    if opt.lr_sch == 'multi_step':
        return lr_scheduler.MultiStepLR(optimizer, milestones=steps, gamma=opt.lr_decay)
    if opt.lr_sch == 'exp':
        return lr_scheduler.ExponentialLR(optimizer, gamma=opt.lr_decay)
    if opt.lr_sch == 'auto':
        return lr_scheduler.ReduceLROnPlateau(optimizer, factor=opt.lr_decay, patience=steps[0])
    return lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda epoch: opt.lr_decay ** (epoch / steps[0]))


def step_scheduler(opt, sch, eval_error_f):
    """train_dual.py:261-264: the plateau schedule watches the evaluation normal error."""
    if opt.lr_sch == 'auto':
        sch.step(eval_error_f)
    else:
        sch.step()


class EvalMeter(object):
    """Node-count-weighted means of the evaluation pass (train_dual.py:233-259): losses and errors of each mesh
    weighted by its vertex / face count, summed -- across ranks too (six scalars, one all-reduce) -- then divided."""

    def __init__(self):
        self.sums = [0.0] * 6         # loss_v * nv, loss_f * nf, error_v * nv, error_f * nf, nv, nf

    def add(self, loss_v, loss_f, error_v, error_f, num_v, num_f):
        for i, v in enumerate((float(loss_v) * num_v, float(loss_f) * num_f, float(error_v) * num_v,
                               float(error_f) * num_f, num_v, num_f)):
            self.sums[i] += v

    def add_prediction(self, vert_p, norm_p, data_v, data_f, loss_v='L1', loss_n='L1'):
        self.add(network.loss_v(vert_p, data_v.y, loss_v), network.loss_n(norm_p, data_f.y, loss_n),
                 network.error_v(vert_p, data_v.y), network.error_n(norm_p, data_f.y),
                 data_v.y.shape[0], data_f.y.shape[0])

    def result(self, device=None):
        """dict(eval_loss_v, eval_loss_f, eval_error_v, eval_error_f); reduced over the process group if there is one."""
        s = self.sums if device is None else reduce_sums(self.sums, device)
        cv, cf = max(s[4], 1.0), max(s[5], 1.0)
        return {'eval_loss_v': s[0] / cv, 'eval_loss_f': s[1] / cf, 'eval_error_v': s[2] / cv, 'eval_error_f': s[3] / cf}


class BestCheckpoint(object):
    """train_dual.py:270-276: keep the state dict with the lowest evaluation normal error.  Keys are the module's own
    (= the reference's: gnn_v.l_conv1.lin.weight ... fc_f2.bias), values plain tensors that
    torch.load(weights_only=True) and the reference's net.load_state_dict(torch.load(path)) both read."""

    def __init__(self, path):
        self.path, self.best = path, math.inf

    def update(self, net, eval_error_f):
        if not (eval_error_f < self.best):
            return False
        self.best = float(eval_error_f)
        if self.path:
            torch.save({k: v.detach().cpu().clone() for k, v in net.state_dict().items()}, self.path)
        return True


# ------------------------------------------------------------------------------------------------ TensorBoard scalars
# /root/reference/code/train_dual.py:134-136,222-226,262-266 logs through tensorboardX.SummaryWriter (add_scalar /
# add_text).  tensorboardX is not a dependency here: the event-file format is small enough to write directly -- TFRecord
# framing (length, masked CRC32-C of the length, payload, masked CRC32-C of the payload) around hand-encoded `Event` protobufs
# (wall_time = 1: double, step = 2: varint, file_version = 3: string, summary = 5: { value = 1: { tag = 1: string,
# simple_value = 2: float } }).  TensorBoard reads the files as it reads tensorboardX's.
def _crc32c_table():
    tab = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        tab.append(c)
    return tab


_CRC_TAB = _crc32c_table()


def _crc32c(data):
    c = 0xFFFFFFFF
    for b in data:
        c = _CRC_TAB[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _masked_crc(data):
    c = _crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(v):
    out = bytearray()
    v &= (1 << 64) - 1
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _field_bytes(num, payload):
    return _varint((num << 3) | 2) + _varint(len(payload)) + payload


class SummaryWriter(object):
    """The subset of tensorboardX.SummaryWriter the reference's driver uses: ``add_scalar(tag, value, step)``,
    ``add_text(tag, text)`` (kept as a ``<tag>.txt`` file beside the events: the text plugin's tensor encoding is not
    reproduced), ``flush()``, ``close()``; one ``events.out.tfevents.<time>.<host>`` file per writer in ``log_dir``."""

    def __init__(self, log_dir):
        import os, socket, struct, time
        self._struct, self._time = struct, time
        os.makedirs(log_dir, exist_ok=True)
        self.log_dir = log_dir
        self.path = os.path.join(log_dir, 'events.out.tfevents.%010d.%s' % (int(time.time()), socket.gethostname()))
        self._f = open(self.path, 'wb')
        self._event(_field_bytes(3, b'brain.Event:2'), step=None)

    def _event(self, body, step):
        s = self._struct
        ev = b'\x09' + s.pack('<d', self._time.time())
        if step is not None:
            ev += b'\x10' + _varint(int(step))
        ev += body
        head = s.pack('<Q', len(ev))
        self._f.write(head + s.pack('<I', _masked_crc(head)) + ev + s.pack('<I', _masked_crc(ev)))

    def add_scalar(self, tag, value, global_step=0):
        val = _field_bytes(1, str(tag).encode('utf-8')) + b'\x15' + self._struct.pack('<f', float(value))
        self._event(_field_bytes(5, _field_bytes(1, val)), global_step)

    def add_text(self, tag, text, global_step=0):
        import os
        with open(os.path.join(self.log_dir, '%s.txt' % str(tag).replace('/', '_')), 'w') as f:
            f.write(str(text))

    def flush(self):
        self._f.flush()

    def close(self):
        if not self._f.closed:
            self._f.close()


def read_scalars(path):
    """[(step, tag, value)] of an event file written by SummaryWriter (or tensorboardX): checks the record CRCs and decodes
    the scalar summaries -- the reader half of the format, for tests and for tools that plot without TensorBoard."""
    import struct

    def varint(buf, i):
        v, sh = 0, 0
        while True:
            b = buf[i]; i += 1
            v |= (b & 0x7F) << sh; sh += 7
            if not b & 0x80:
                return v, i

    def fields(buf):
        i = 0
        while i < len(buf):
            key, i = varint(buf, i)
            num, wt = key >> 3, key & 7
            if wt == 0:
                v, i = varint(buf, i)
            elif wt == 1:
                v = buf[i:i + 8]; i += 8
            elif wt == 5:
                v = buf[i:i + 4]; i += 4
            elif wt == 2:
                n, i = varint(buf, i)
                v = buf[i:i + n]; i += n
            else:
                raise ValueError('wire type %d' % wt)
            yield num, wt, v

    out = []
    data = open(path, 'rb').read()
    i = 0
    while i < len(data):
        head = data[i:i + 8]
        n, = struct.unpack('<Q', head)
        if struct.unpack('<I', data[i + 8:i + 12])[0] != _masked_crc(head):
            raise ValueError('length CRC mismatch at byte %d' % i)
        ev = data[i + 12:i + 12 + n]
        if struct.unpack('<I', data[i + 12 + n:i + 16 + n])[0] != _masked_crc(ev):
            raise ValueError('payload CRC mismatch at byte %d' % i)
        i += 16 + n
        step, summary = 0, None
        for num, wt, v in fields(ev):
            if num == 2 and wt == 0:
                step = v
            elif num == 5 and wt == 2:
                summary = v
        if summary is None:
            continue
        for num, wt, v in fields(summary):
            if num != 1:
                continue
            tag, val = None, None
            for n2, w2, v2 in fields(v):
                if n2 == 1:
                    tag = bytes(v2).decode('utf-8')
                elif n2 == 2 and w2 == 5:
                    val, = struct.unpack('<f', v2)
            if tag is not None and val is not None:
                out.append((step, tag, val))
    return out

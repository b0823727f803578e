"""Data parallelism for the mesh path: one process per GPU, meshes sharded, gradients all-reduced.

The reference is single-device; its "batch_size" is gradient accumulation over sequential
single-mesh steps (/root/reference/code/train_dual.py:211-218).  Meshes are independent units
(no graph is ever split), so the only collective is ONE all-reduce of the flat fp32 gradient
bucket (939 128 elements = 3.76 MB) per optimiser step -- RCCL over xGMI with backend "nccl",
gloo on CPU for tests.  Eval metrics reduce as (sum, count) pairs like train_dual.py:246-259.
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run); returns (rank, world, device)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    use_cuda = torch.cuda.is_available()
    if use_cuda and os.environ.get('GEOBI_ALL_RANKS_ON_DEVICE0') == '1':
        local = 0          # rehearsal of the multi-rank path on a 1-GPU box (use with GEOBI_DIST_BACKEND=gloo)
    device = torch.device('cuda', local) if use_cuda else torch.device('cpu')
    if use_cuda:
        torch.cuda.set_device(device)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        backend = backend or os.environ.get('GEOBI_DIST_BACKEND') or ('nccl' if use_cuda else 'gloo')
        kw = {}
        if use_cuda and backend == 'nccl':
            kw['device_id'] = device
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, device


def shard_indices(num_items, rank, world, seed=0, epoch=0, shuffle=True):
    """Round-robin shard of a common permutation (same seed on every rank); every item appears
    on exactly one rank, ranks differ in length by at most one."""
    if shuffle:
        g = torch.Generator().manual_seed(seed * 1000003 + epoch)
        order = torch.randperm(num_items, generator=g).tolist()
    else:
        order = list(range(num_items))
    return order[rank::world]


def rank_world():
    """(rank, world size) of the default process group, (0, 1) without one."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def owns_patch(k, rank, world):
    """Inference on one large mesh across GPUs: patch k of the split (test_dual.py:53-58 runs them one
    after the other) belongs to rank k mod world."""
    return k % world == rank


def reduce_patch_sums(tensors, dst=0):
    """Sum the per-rank patch accumulators (Vp [V,3], Np [F,3], visit counts [V]) onto rank `dst`, in place there:
    the one collective of multi-GPU inference.  The buffers are packed into one fp32 message (counts are small
    integers, exact in fp32) so a mesh costs one reduction, not three."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return tensors
    flat = torch.cat([t.reshape(-1).to(torch.float32) for t in tensors])
    if flat.is_cuda and dist.get_backend() == 'gloo':
        # rehearsal of the multi-rank path on one device (GEOBI_DIST_BACKEND=gloo): gloo has no device-side reduce
        host = flat.cpu()
        dist.reduce(host, dst=dst, op=dist.ReduceOp.SUM)
        flat = host.to(flat.device)
    else:
        dist.reduce(flat, dst=dst, op=dist.ReduceOp.SUM)       # RCCL over xGMI
    if dist.get_rank() == dst:
        off = 0
        for t in tensors:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t).to(t.dtype))
            off += n
    return tensors


class GradBucket(object):
    """All parameter gradients as views of one flat fp32 buffer: zeroing is one memset, the
    data-parallel reduction is one all-reduce, and nothing is copied in or out."""

    def __init__(self, params, direct=False):
        """direct=True lets the backward kernels ADD gradients straight into the bucket views
        (ops._direct_grad, `accumulate` in the C ABI): same semantics as autograd's accumulation into
        ``.grad`` -- backward passes between two `zero()` calls sum up -- without the per-parameter
        accumulate kernels."""
        self.params = [p for p in params if p.requires_grad]
        total = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(total, dtype=ref.dtype, device=ref.device)
        self.owner = None            # FlatParameters: the flat Parameter whose .grad must stay `self.flat`
        self._views = []
        off = 0
        for p in self.params:
            n = p.numel()
            v = self.flat[off:off + n].view_as(p)
            self._views.append(v)
            p.grad = v
            p._geobi_direct_grad = bool(direct)
            p._geobi_grad_ptr = v.data_ptr()
            p._geobi_bucket = self           # ops._direct_grad also checks the flat owner's .grad (FlatParameters)
            off += n

    def zero(self):
        """Zero every gradient (one memset) and re-attach any view an optimizer's
        ``zero_grad(set_to_none=True)`` or a stray assignment dropped since the last call."""
        self.flat.zero_()
        for p, v in zip(self.params, self._views):
            if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                p.grad = v
        if self.owner is not None and self.owner.grad is not self.flat:
            self.owner.grad = self.flat

    def facet_offset(self, net):
        """Element offset in `flat` where the facet half (gnn_f, fc_f1, fc_f2) starts: DualGNN registers gnn_v, fc_v1,
        fc_v2 first, so the halves are two contiguous ranges.  Its gradients are final BEFORE the vertex branch's
        backward runs (the backward walks facet head -> gnn_f -> coupling -> vertex head -> gnn_v)."""
        first = next(net.gnn_f.parameters())
        off = 0
        for p in self.params:
            if p is first:
                return off
            off += p.numel()
        raise ValueError('the bucket does not hold the network\'s parameters')

    def all_reduce_mean_split(self, offset, events=None, comm_stream=None, _single_rank_too=False):
        """The same reduction as `all_reduce_mean` in two collectives: the tail [offset:] (the facet half) FIRST -- on
        `comm_stream` behind `events` (torch.cuda.Event list recorded by geobi_net_backward_facet_events) when given, so
        that it runs under the vertex branch's backward -- then the head [:offset].  Element-wise the same sums, so the
        reduced bucket is identical to the one-shot all-reduce."""
        if not (dist.is_available() and dist.is_initialized()):
            return self.flat
        if dist.get_world_size() == 1 and not _single_rank_too:      # (tests run the collectives on one rank as well)
            return self.flat
        world = dist.get_world_size()
        tail, head = self.flat[offset:], self.flat[:offset]
        # the early collective needs a backend that reduces on the device (RCCL); gloo stages through the host and only
        # makes sense one collective after the other (the rehearsal / CPU path)
        if events and self.flat.is_cuda and comm_stream is not None and dist.get_backend() == 'nccl':
            for ev in events:
                comm_stream.wait_event(ev)
            with torch.cuda.stream(comm_stream):
                work = dist.all_reduce(tail, op=dist.ReduceOp.SUM, async_op=True)
            dist.all_reduce(head, op=dist.ReduceOp.SUM)          # behind the whole backward, on the compute stream
            work.wait()                                          # the compute stream waits for the early collective
        else:
            dist.all_reduce(tail, op=dist.ReduceOp.SUM)
            dist.all_reduce(head, op=dist.ReduceOp.SUM)
        self.flat.div_(world)
        return self.flat

    def all_reduce_mean(self):
        """Sum over ranks then divide by the world size (= the reference's loss / batch_size)."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.flat.div_(dist.get_world_size())
        return self.flat


class FlatParameters(object):
    """Parameters AND gradients as views of two flat buffers.

    The optimiser then updates ONE tensor (`flat_param`, 939 128 elements) instead of 76 small
    ones -- a handful of kernel launches per step -- and the module's own Parameter objects (and
    hence its state_dict) keep working because they alias the flat storage."""

    def __init__(self, module, direct=True):
        params = [p for p in module.parameters() if p.requires_grad]
        flat = torch.cat([p.detach().reshape(-1) for p in params])
        self.flat_param = torch.nn.Parameter(flat)
        off = 0
        for p in params:
            n = p.numel()
            p.data = self.flat_param.data[off:off + n].view_as(p)
            off += n
        self.bucket = GradBucket(params, direct=direct)
        self.bucket.owner = self.flat_param
        self.flat_param.grad = self.bucket.flat

    def parameters(self):
        return [self.flat_param]


def reduce_sums(values, device):
    """All-reduce a list of python/tensor scalars (sums and counts of the eval loop)."""
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if t.is_cuda:
            t = t.float()          # RCCL path: fp32 is plenty for six scalars
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.double().tolist()


def _mesh_weights(data):
    """Per-row weights 1 / (B * n_mesh) of a disjoint-union batch (cached on the Data bag)."""
    ptr = getattr(data, 'mesh_ptr', None)
    if ptr is None or ptr.numel() <= 2:
        return None
    w = getattr(data, '_loss_weights', None)
    if w is None or w.device != data.y.device:
        dev = data.y.device
        counts = ptr[1:] - ptr[:-1]
        per_mesh = (1.0 / (counts.float() * counts.numel())).to(dev)
        if dev.type == 'cuda' and not ptr.is_cuda:
            # expand on the device (two small kernels) instead of on the host + a copy of N floats
            seg = torch.bucketize(torch.arange(int(ptr[-1]), device=dev), ptr[1:].to(dev), right=True)
            w = per_mesh[seg]
        else:
            w = torch.repeat_interleave(per_mesh, counts.to(dev))
        data._loss_weights = w
    return w


def batched_losses(vp, npred, data_v, data_f, loss_v='L1', loss_n='L1'):
    """Per-mesh mean losses averaged over the meshes of a disjoint-union batch.

    Identical to accumulating ``loss / batch_size`` over sequential single-mesh steps
    (train_dual.py:204-212).  Without ``mesh_ptr`` it is the plain per-node mean.  On the MI355X
    this is the fused reduction kernel (geobi_row_loss_*); CPU tensors (gloo tests) use torch ops."""
    kinds = {'L1': 0, 'L2': 1}
    if vp.is_cuda:
        from . import ops
        wv, wn = _mesh_weights(data_v), _mesh_weights(data_f)
        lv = ops.row_loss(vp, data_v.y, kinds[loss_v], wv, 1.0 if wv is not None else None)
        ln = ops.row_loss(npred, data_f.y, kinds[loss_n], wn, 1.0 if wn is not None else None)
        return lv, ln

    def per_node(a, b, kind):
        d = a - b
        return d.abs().sum(1) if kind == 'L1' else d.pow(2).sum(1)

    def reduce(vals, ptr):
        if ptr is None or ptr.numel() <= 2:
            return vals.mean()
        counts = (ptr[1:] - ptr[:-1]).to(vals.device)
        w = torch.repeat_interleave(1.0 / (counts.to(vals.dtype) * counts.numel()), counts)
        return (vals * w).sum()

    lv = reduce(per_node(vp, data_v.y, loss_v), getattr(data_v, 'mesh_ptr', None))
    ln = reduce(per_node(npred, data_f.y, loss_n), getattr(data_f, 'mesh_ptr', None))
    return lv, ln

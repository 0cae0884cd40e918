"""One process per GPU over RCCL — rank discovery and the single metric exchange of the sharded path.

Pattern after reference ``utils/dist.py:12-32`` (env-var rank discovery, backend string ``"nccl"`` —
which is RCCL on ROCm — ``env://`` rendezvous, one barrier) and ``:79-102`` (``reduce_dict``).  The bg
forecasting path shards by sample with no data-path collective; the only exchange is the end-of-run
gather of the [n_cls,4] float64 PQ accumulators (608 B per rank for 19 classes): latency-bound, one
``all_gather_into_tensor``.
"""
import os

import torch
import torch.distributed as dist


def env_rank():
    return (int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1)),
            int(os.environ.get('LOCAL_RANK', 0)))


def init_distributed_mode(backend=None, device=None):
    """Returns (rank, world_size, local_rank).  No-op for a single process."""
    rank, world, local = env_rank()
    if world <= 1:
        return rank, world, local
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29500')
    if backend is None:
        # PF_DIST_BACKEND=gloo: several ranks on ONE GPU (tests on a 1-GPU box); RCCL refuses two ranks per device
        backend = os.environ.get('PF_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
    if backend == 'nccl':
        torch.cuda.set_device(local)
        device = torch.device('cuda', local)
    if not dist.is_initialized():
        kw = {'device_id': device} if backend == 'nccl' and device is not None else {}
        dist.init_process_group(backend=backend, init_method='env://', rank=rank, world_size=world, **kw)
    dist.barrier()
    return rank, world, local


def is_dist():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def shard_indices(n_items, rank, world):
    """Sample partition of the val list: rank r takes r, r+G, r+2G, ...."""
    return list(range(rank, n_items, world))


def _host_collectives():
    """gloo carries the collectives of CPU runs and of several ranks sharing one GPU (tests): its tensors travel through host memory"""
    return str(dist.get_backend()) == 'gloo'


def gather_accumulators(acc):
    """acc [n_cls,4] float64 on this rank -> [world, n_cls, 4] on every rank (identity for 1 process)."""
    if not is_dist():
        return acc.unsqueeze(0)
    if acc.is_cuda and _host_collectives():
        return gather_accumulators(acc.cpu()).to(acc.device)
    world = dist.get_world_size()
    # concatenated along dim 0 (the layout every backend accepts), viewed back as [world, ...]
    out = torch.empty((world * acc.shape[0],) + tuple(acc.shape[1:]), dtype=acc.dtype, device=acc.device)
    dist.all_gather_into_tensor(out, acc.contiguous())
    return out.view((world,) + tuple(acc.shape))


def gather_objects(obj):
    """obj of every rank, in rank order, on every rank ([obj] for 1 process)."""
    if not is_dist():
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


def device_identity(local):
    """A string that differs between the physical GPUs of a node (uuid when the runtime reports one, else the PCI address);
    a CPU-only process (gloo dry runs) reports its rank's host slot."""
    if not torch.cuda.is_available():
        return 'cpu:%d' % local
    p = torch.cuda.get_device_properties(local)
    uuid = getattr(p, 'uuid', None)
    if uuid is not None and str(uuid).strip('0-') != '':
        return str(uuid)
    return 'pci:%s:%s:%s' % (getattr(p, 'pci_domain_id', '?'), getattr(p, 'pci_bus_id', '?'), getattr(p, 'pci_device_id', local))


def backend_description():
    """What carries the collectives of this run: ``'single process'``, or the torch.distributed backend string with the
    RCCL version when it is "nccl" (which is RCCL on ROCm)."""
    if not is_dist():
        return 'single process'
    b = str(dist.get_backend())
    if b == 'nccl':
        try:
            v = torch.cuda.nccl.version()
            v = '.'.join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
        except Exception as e:   # pragma: no cover - version probe only
            v = 'unknown (%s)' % type(e).__name__
        return 'nccl = RCCL %s, world %d' % (v, dist.get_world_size())
    return '%s, world %d' % (b, dist.get_world_size())


def max_over_ranks(x, device):
    if not is_dist():
        return float(x)
    t = torch.tensor([float(x)], dtype=torch.float64, device='cpu' if _host_collectives() else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_reduce_mean_(flat):
    """Data-parallel gradient exchange of the bg training step: ONE sum all-reduce of the flat gradient arena (16.5 MB
    for FC-HarDNet-70) and a division by the world size — what DistributedDataParallel does bucket by bucket in the
    reference (``training/train.py:96-103``).  In place; identity for a single process."""
    if is_dist():
        if flat.is_cuda and _host_collectives():
            host = flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            flat.copy_(host)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(dist.get_world_size())
    return flat

"""Pair-sharded multi-GPU evaluation of the Gaussian-distance loss (one process per GPU).

Box pairs are independent, so the batch shards as contiguous row ranges with NO data-path
collective: every rank runs the fused kernel on its own rows and already holds the final
gradient of its rows (the global loss is the sum of the shard losses).  The only exchange is one
fp32 per rank: an ``all_gather`` of the per-shard losses (RCCL over xGMI on MI355X: backend
``"nccl"``), summed in rank order so every rank gets the bit-identical global value.  It is
latency-bound (32 B on 8 GPUs), so it is issued asynchronously and joined as late as the caller
allows; the kernel of the next step overlaps it.

In the reference this role is played by DDP + per-rank batches (tools/dist_train.sh:8-9,
tools/train.py:130-137); the loss scalar itself is never all-reduced there — the collective here
exists to report the global loss (logging / the benchmark's parity check).
"""
import torch
import torch.distributed as dist


def shard_range(n, rank, world_size):
    """Contiguous row range [lo, hi) of `n` pairs owned by `rank` (balanced to within one row)."""
    if not (0 <= rank < world_size):
        raise ValueError(f'rank {rank} outside world of {world_size}')
    lo = (n * rank) // world_size
    hi = (n * (rank + 1)) // world_size
    return lo, hi


class _GatherSum(torch.autograd.Function):
    """all_gather of one scalar per rank, summed in rank order.  d(global)/d(local) = 1 on every rank."""

    @staticmethod
    def forward(ctx, local, group):
        world = dist.get_world_size(group)
        parts = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(parts, local.contiguous(), group=group)
        out = parts[0].clone()
        for p in parts[1:]:
            out = out + p  # fixed order: identical bits on every rank
        return out

    @staticmethod
    def backward(ctx, grad_out):
        return grad_out, None


class PendingGather:
    """Handle of an in-flight all_gather of shard losses; `.result()` joins it."""

    def __init__(self, gathered, work):
        self._gathered = gathered   # (world, ...) — row r = rank r's shard loss(es)
        self._work = work

    def result(self):
        if self._work is not None:
            self._work.wait()  # stream-ordered on GPU backends: the host does not block
            self._work = None
        shard_losses = self._gathered
        total = shard_losses[0].clone()
        for r in range(1, shard_losses.shape[0]):
            total = total + shard_losses[r]  # fixed rank order: identical bits on every rank
        return total, shard_losses


def gather_shard_losses(local_loss, group=None, async_op=True):
    """Start the all_gather of this rank's (detached) shard loss — a scalar, or a small vector when several losses
    share one collective.  Returns a PendingGather whose result() is (sum over ranks, per-rank stack)."""
    if not (dist.is_available() and dist.is_initialized()):
        return PendingGather(local_loss.detach().clone().unsqueeze(0), None)
    world = dist.get_world_size(group)
    local = local_loss.detach().contiguous()
    # one output tensor, one collective kernel (the list form of all_gather adds `world` copy kernels per call)
    flat = local.reshape(-1)    # 1-D in, (world * k) out: the layout every backend accepts (gloo rejects 0-dim inputs)
    gathered = torch.empty(world * flat.numel(), dtype=local.dtype, device=local.device)
    work = dist.all_gather_into_tensor(gathered, flat, group=group, async_op=async_op)
    return PendingGather(gathered.view((world,) + tuple(local.shape)), work if async_op else None)


class ShardedGDLoss(torch.nn.Module):
    """Global mean/sum Gaussian-distance loss over pairs sharded across ranks.

    loss_module : a GDLoss (or any callable with its forward signature) built with
                  reduction 'mean' or 'sum'.
    forward(pred_shard, target_shard, weight_shard=None, total_pairs=None, avg_factor=None)
        pred_shard/target_shard: this rank's rows.  With reduction 'mean' the normaliser is the
        GLOBAL pair count (`total_pairs`, default: sum of shard sizes via one extra all_reduce)
        unless `avg_factor` is given.  Returns the differentiable global loss; backward leaves the
        final gradient of this rank's rows in pred_shard.grad (no gradient exchange is needed).
    """

    def __init__(self, loss_module, group=None):
        super().__init__()
        self.loss_module = loss_module
        self.group = group

    def local_loss(self, pred, target, weight=None, total_pairs=None, avg_factor=None):
        reduction = getattr(self.loss_module, 'reduction', 'mean')
        if reduction == 'none':
            raise ValueError('ShardedGDLoss needs a reduced loss (mean or sum)')
        if reduction == 'mean' and avg_factor is None:
            if total_pairs is None:
                cnt = torch.tensor([pred.reshape(-1, 7).shape[0]], dtype=torch.int64, device=pred.device)
                if dist.is_available() and dist.is_initialized():
                    dist.all_reduce(cnt, group=self.group)
                total_pairs = int(cnt.item())
            avg_factor = total_pairs
        if reduction == 'sum':
            return self.loss_module(pred, target, weight)
        return self.loss_module(pred, target, weight, avg_factor=avg_factor)

    def forward(self, pred, target, weight=None, total_pairs=None, avg_factor=None):
        local = self.local_loss(pred, target, weight, total_pairs, avg_factor)
        if not (dist.is_available() and dist.is_initialized()):
            return local
        return _GatherSum.apply(local, self.group)

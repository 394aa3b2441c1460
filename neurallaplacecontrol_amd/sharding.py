"""Host-side logic of the K-sharded planner: which samples a rank owns and the one
collective per command().  Kept free of GPU calls so the world_size > 1 path is testable with gloo on CPU."""

import torch


def shard_range(K, world_size, rank):
    """Rank r of G owns global samples [r*K/G, (r+1)*K/G); K must divide evenly (all BASELINE configs do)."""
    if K % world_size:
        raise ValueError(f"num_samples={K} must be divisible by the group size {world_size}")
    k_local = K // world_size
    return rank * k_local, k_local


def partial_width(T, nu):
    """(beta_r, eta_r, S_r[t, j]) -> 2 + T*nu doubles per rank (336 B at T=40, nu=1)."""
    return 2 + T * nu


def gather_partials(partials, gathered, group):
    """One all-gather of every rank's (2 + T*nu) partial vector; RCCL over xGMI on GPUs, gloo in CPU tests."""
    import torch.distributed as dist

    assert gathered.numel() == dist.get_world_size(group) * partials.numel()
    if partials.is_cuda and dist.get_backend(group) == "gloo":
        # gloo has no device all-gather: stage through the host (tests that run two ranks on ONE GPU; RCCL refuses
        # two ranks per device)
        host = torch.empty(gathered.shape, dtype=gathered.dtype)
        dist.all_gather_into_tensor(host.view(-1), partials.detach().cpu().contiguous().view(-1), group=group)
        gathered.copy_(host)
        return gathered
    dist.all_gather_into_tensor(gathered.view(-1), partials.contiguous().view(-1), group=group)
    return gathered


def merge_partials_torch(gathered, lambda_):
    """The shard merge of SURVEY 8e as tensor ops (the planners the HIP kernels are not built for, MPPIDelay._torch_command;
    merge_kernel does the same on the HIP path): rows (beta_r, eta_r, S_r[t, j]) of every rank -> beta = min beta_r,
    scale_r = exp(-(beta_r - beta) / lambda), eta = sum scale_r eta_r, S = sum scale_r S_r.  Returns (beta, eta, S)."""
    beta_r, eta_r, S_r = gathered[:, 0], gathered[:, 1], gathered[:, 2:]
    beta = beta_r.min()
    scale = torch.exp(-(beta_r - beta) / lambda_)
    return beta, (scale * eta_r).sum(), (scale.view(-1, 1) * S_r).sum(dim=0)


def slice_noise(raw, k_offset, k_local):
    """Every rank draws the SAME (K, T, nu) tensor from the same seed and keeps its slice, so the sharded
    run consumes the torch generator exactly like the single-GPU / reference run."""
    return raw[k_offset : k_offset + k_local]


def replicate_from_rank0(t, group, compute_device=None):
    """Every rank of a K-sharded planner must hold the SAME control sequence U (merge_kernel applies the same update to
    each rank's own copy): U is drawn by the host RNG in the constructor / reset(), so instead of trusting identical
    seeds the value of rank 0 is broadcast.  Returns a tensor like `t` (host float64)."""
    import torch.distributed as dist

    out = t.detach().to("cpu", torch.float64).contiguous().clone()
    if dist.get_backend(group) == "gloo" or compute_device is None:
        dist.broadcast(out, src=dist.get_global_rank(group, 0), group=group)
        return out
    dev = out.to(compute_device)  # RCCL moves device memory
    dist.broadcast(dev, src=dist.get_global_rank(group, 0), group=group)
    return dev.cpu()


def check_same_on_all_ranks(values, group, what, compute_device=None):
    """Raise if a small tuple of numbers (seeds, sizes) differs between the ranks of `group`."""
    import torch.distributed as dist

    mine = torch.tensor([float(v) for v in values], dtype=torch.float64)
    G = dist.get_world_size(group)
    if dist.get_backend(group) != "gloo" and compute_device is not None:
        mine = mine.to(compute_device)
    allv = torch.empty(G * mine.numel(), dtype=torch.float64, device=mine.device)
    dist.all_gather_into_tensor(allv, mine, group=group)
    allv = allv.view(G, -1).cpu()
    if not bool((allv == allv[0]).all()):
        raise ValueError(f"K-sharded planner: {what} differ between ranks: {allv.tolist()}")


def share_bytes_from_rank0(payload, nbytes, group, compute_device=None):
    """Rank 0's `payload` (bytes of length nbytes; ignored elsewhere) on every rank of `group` -- the unique id of the
    library's own communicator (include/nlc.h, nlc_comm_init) travels through the group the caller already has."""
    import torch.distributed as dist

    t = torch.zeros(nbytes, dtype=torch.uint8)
    if dist.get_rank(group) == 0:
        t = torch.frombuffer(bytearray(payload), dtype=torch.uint8).clone()
    if dist.get_backend(group) != "gloo" and compute_device is not None:
        t = t.to(compute_device)
    dist.broadcast(t, src=dist.get_global_rank(group, 0), group=group)
    return bytes(t.cpu().tolist())


def all_ranks_agree(flag, group, compute_device=None):
    """True iff `flag` is truthy on EVERY rank of `group` (a min-all-reduce of one number)."""
    import torch.distributed as dist

    t = torch.tensor([1.0 if flag else 0.0], dtype=torch.float64)
    if dist.get_backend(group) != "gloo" and compute_device is not None:
        t = t.to(compute_device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(t.cpu().item() > 0.5)
